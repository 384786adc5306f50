#!/usr/bin/env python3
"""Drop-in entry point with the reference's file name and command line (vclust.py v1.3.1):
`vclust.py prefilter|align|cluster|deduplicate|info ...`.  See vclust_amd/cli.py."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))

from vclust_amd.cli import ALIGN_FIELDS, ALIGN_OUTFMT, __version__, main  # noqa: E402,F401

if __name__ == '__main__':
    main()
