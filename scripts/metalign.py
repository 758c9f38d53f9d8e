#! /usr/bin/env python
"""Entry point with the reference's script name (/root/reference/scripts/metalign.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import metalign as _m  # noqa: E402

if __name__ == '__main__':
    _m.main()
