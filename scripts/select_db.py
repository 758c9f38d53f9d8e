#! /usr/bin/env python
"""Entry point with the reference's script name (/root/reference/scripts/select_db.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import select_db as _m  # noqa: E402

if __name__ == '__main__':
    _m.select_main(_m.select_parseargs())
