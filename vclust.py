#!/usr/bin/env python3
"""Drop-in entry point with the reference's file name and command line (vclust.py v1.3.1):
`vclust.py prefilter|align|cluster|deduplicate|info ...`.  See vclust_amd/cli.py."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

from vclust_amd.cli import ALIGN_FIELDS, ALIGN_OUTFMT, __version__, main  # noqa: E402,F401

if __name__ == '__main__':
    import os
    rc = 0
    try:
        main()
    except SystemExit as exc:          # argparse / handlers: same codes as the reference (0, 1, 2)
        rc = exc.code if isinstance(exc.code, int) else (0 if exc.code is None else 1)
        if exc.code is not None and not isinstance(exc.code, int):
            print(exc.code, file=sys.stderr)
    # a one-shot process: everything it wrote is closed; skip the interpreter's and the HIP runtime's tear-down
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(rc)
