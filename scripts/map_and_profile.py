#! /usr/bin/env python
"""Entry point with the reference's script name (/root/reference/scripts/map_and_profile.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import map_and_profile as _m  # noqa: E402

if __name__ == '__main__':
    _m.map_main(_m.profile_parseargs())
