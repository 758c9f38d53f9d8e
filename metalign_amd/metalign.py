#! /usr/bin/env python
"""Pipeline driver: pre-filter the database on the MI355X, align, profile.

Takes the command line of /root/reference/scripts/metalign.py (flags declared in metalign_amd/cli.py) and hands
one mutable Namespace to the two stages, as the reference does (:77-85): `select_db.select_main` writes the
subset database and its db_info into temp_dir, `map_and_profile.map_main` aligns the reads against it and
writes the CAMI profile.
"""
import shutil
import sys
import tempfile

from . import cli
from . import map_and_profile as mapper
from . import select_db as select


def metalign_parseargs(argv=None):
    return cli.parser_for('metalign').parse_args(argv)


def _apply_modes(args):
    """--sensitive / --precise rewrite other options (reference :68-74)."""
    if args.sensitive and args.precise:
        sys.exit('You cannot use both --sensitive and --precise.')
    if args.sensitive:
        args.cutoff = 0.0
    if args.precise:
        args.read_cutoff, args.min_abundance = 100, 0.1


def _wire_stages(args):
    """Fill in the options the two stages read but this command line does not expose (reference :77-81)."""
    args.db = args.temp_dir + 'cmashed_db.fna'
    args.dbinfo = args.dbinfo_out = args.temp_dir + 'subset_db_info.txt'
    args.infiles = [args.reads]
    args.cmash_results = 'NONE'


def main(argv=None):
    args = metalign_parseargs(argv)
    args.data = cli.with_slash(args.data)
    own_temp = args.temp_dir == 'AUTO/'  # a directory of this process's own making (one per rank in a multi-GPU launch)
    if own_temp:
        args.temp_dir = tempfile.mkdtemp(prefix=args.data)
    args.temp_dir = cli.with_slash(args.temp_dir)
    if args.dbinfo_in == 'AUTO':
        args.dbinfo_in = args.data + 'db_info.txt'
    if args.db_dir == 'AUTO':
        args.db_dir = args.data + 'organism_files/'
    if args.input_type == 'AUTO':
        args.input_type = cli.sniff_reads_type(args.reads)
    _apply_modes(args)
    _wire_stages(args)
    select.select_main(args)
    import os
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and int(os.environ.get('RANK', '0')) != 0:
        # a multi-GPU launch (torch.distributed.run): the ranks share stages A + B; rank 0 aligns and profiles.
        # An explicit --temp_dir is ONE directory for all ranks: rank 0 is still reading the CSV from it, writing the
        # subset database into it and aligning against it — only rank 0 removes it, after map_main.  A rank's own
        # mkdtemp (AUTO) holds nothing anybody else reads.
        if own_temp and not args.keep_temp_files:
            shutil.rmtree(args.temp_dir, ignore_errors=True)
        return
    mapper.map_main(args)
    if not args.keep_temp_files:
        shutil.rmtree(args.temp_dir, ignore_errors=True)


if __name__ == '__main__':
    main()
