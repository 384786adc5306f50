#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <chrono>
bool vg_fast_gunzip(const unsigned char* in, size_t n, int n_threads, char** out_p, size_t* out_n);
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> v((size_t)n); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) return 3; fclose(f);
    char* o = nullptr; size_t on = 0;
    auto t0 = std::chrono::steady_clock::now();
    bool ok = vg_fast_gunzip(v.data(), v.size(), 8, &o, &on);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) { printf("REFUSED\n"); return 2; }
    fprintf(stderr, "%.1f MB in %.3f s = %.0f MB/s\n", on / 1e6, dt, on / 1e6 / dt);
    if (argc > 2) { FILE* g = fopen(argv[2], "wb"); fwrite(o, 1, on, g); fclose(g); }
    printf("OK %zu\n", on); free(o); return 0;
}
