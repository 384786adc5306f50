// Developer micro-benchmark: cost of getting a page-cache resident file in front of T threads --
// mmap + MADV_POPULATE_READ + scan + munmap against pread into per-thread buffers + scan.
// g++ -O2 -pthread -o /tmp/read_cost tools/micro/read_cost.cpp && /tmp/read_cost FILE [threads]
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* path = argv[1]; const int T = argc > 2 ? atoi(argv[2]) : 64;
    int fd = open(path, O_RDONLY); struct stat st; fstat(fd, &st); const size_t n = (size_t)st.st_size;
    const size_t piece = 4u << 20; const size_t n_pieces = (n + piece - 1) / piece;
    for (int rep = 0; rep < 2; ++rep) {
        {   // mmap
            double t0 = now();
            char* m = (char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            std::atomic<size_t> next(0); std::atomic<uint64_t> total(0);
            std::vector<std::thread> th;
            double t1 = now();
            for (int t = 0; t < T; ++t) th.emplace_back([&] {
                uint64_t c = 0;
                for (;;) { size_t i = next.fetch_add(1); if (i >= n_pieces) break; size_t a = i * piece, b = std::min(n, a + piece);
                    madvise(m + a, b - a, MADV_POPULATE_READ);
                    for (const char* p = m + a; (p = (const char*)memchr(p, '>', m + b - p)); ++p) ++c; }
                total += c; });
            for (auto& x : th) x.join();
            double t2 = now();
            munmap(m, n);
            double t3 = now();
            printf("mmap : map %.3f  populate+scan %.3f  munmap %.3f  total %.3f s  ('>' %llu)\n", t1 - t0, t2 - t1, t3 - t2, t3 - t0, (unsigned long long)total.load());
        }
        {   // pread
            double t0 = now();
            std::atomic<size_t> next(0); std::atomic<uint64_t> total(0);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&] {
                std::vector<char> buf(piece); uint64_t c = 0;
                for (;;) { size_t i = next.fetch_add(1); if (i >= n_pieces) break; size_t a = i * piece, b = std::min(n, a + piece);
                    size_t got = 0; while (got < b - a) { ssize_t r = pread(fd, buf.data() + got, b - a - got, (off_t)(a + got)); if (r <= 0) break; got += (size_t)r; }
                    for (const char* p = buf.data(); (p = (const char*)memchr(p, '>', buf.data() + got - p)); ++p) ++c; }
                total += c; });
            for (auto& x : th) x.join();
            double t1 = now();
            printf("pread: read+scan %.3f s  ('>' %llu)\n", t1 - t0, (unsigned long long)total.load());
        }
    }
    return 0;
}
