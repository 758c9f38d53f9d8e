#! /usr/bin/env python
"""Full pipeline driver (drop-in for /root/reference/scripts/metalign.py): same flags, defaults and
'AUTO' / 'NONE' sentinels (:8-40), same --sensitive / --precise rewrites (:68-74), same hand-off of one
mutable Namespace to the two stages (:77-85)."""
import argparse
import subprocess
import sys
import tempfile

from . import map_and_profile as mapper
from . import select_db as select


def metalign_parseargs(argv=None):
    p = argparse.ArgumentParser(description='Runs full metalign pipeline on input reads file(s).')
    p.add_argument('reads', help='Path to reads file.')
    p.add_argument('data', help='Path to data/ directory with the files from setup_data.sh')
    p.add_argument('--cutoff', type=float, default=0.01, help='CMash cutoff value. Default is 0.01.')
    p.add_argument('--db_dir', default='AUTO', help='Directory with all organism files in the full database.')
    p.add_argument('--dbinfo_in', default='AUTO', help='Location of db_info file. Default: data/db_info.txt')
    p.add_argument('--keep_temp_files', action='store_true', help='Retain KMC files after this script finishes.')
    p.add_argument('--input_type', default='AUTO', choices=['fastq', 'fasta', 'AUTO'],
                   help='Type of input file (fastq/fasta). Default: try to auto-determine')
    p.add_argument('--length_normalize', action='store_true', help='Normalize abundances by genome length.')
    p.add_argument('--low_mem', action='store_true',
                   help='Run in low memory mode, with inexact multimapped processing.')
    p.add_argument('--min_abundance', type=float, default=10**-4,
                   help='Minimum abundance for a taxa to be included in the results. Default: 10^(-4).')
    p.add_argument('--no_quantify_unmapped', action='store_true',
                   help='Do not factor in unmapped reads in abundance estimation.')
    p.add_argument('--output', default='abundances.tsv', help='Output abundances file. Default: abundances.tsv')
    p.add_argument('--pct_id', type=float, default=0.5,
                   help='Minimum percent identity from reference to count a hit.')
    p.add_argument('--precise', action='store_true',
                   help='Run in precise mode. Overwrites --read_cutoff and --min_abundance to 100 and 0.1.')
    p.add_argument('--rank_renormalize', action='store_true',
                   help='Renormalize abundances to 100 pct. at each rank, e.g if an organism has a species but not genus label.')
    p.add_argument('--read_cutoff', type=int, default=1, help='Number of reads to count an organism as present.')
    p.add_argument('--sampleID', default='NONE', help='Sample ID for output. Defaults to input file name(s).')
    p.add_argument('--sensitive', action='store_true', help='Run in sensitive mode. Sets --cutoff value to 0.0.')
    p.add_argument('--strain_level', action='store_true', help='Profile strains (off by default).')
    p.add_argument('--temp_dir', default='AUTO/', help='Directory to write temporary files to.')
    p.add_argument('--threads', type=int, default=4, help='Number of compute threads for Minimap2/KMC. Default: 4')
    p.add_argument('--verbose', action='store_true', help='Print verbose output.')
    # build-only additions (defaults keep the reference behaviour)
    p.add_argument('--sketch_table', default='AUTO', help='Genome sketch table directory. Default: data/sketch_table')
    p.add_argument('--min_count', type=int, default=2, help='k-mer count threshold (kmc -ci). Default: 2')
    p.add_argument('--sketch_size', type=int, default=0, help='Read sketch size per k; 0 = every hash <= table max.')
    return p.parse_args(argv)


def main(argv=None):
    args = metalign_parseargs(argv)
    if not args.data.endswith('/'):
        args.data += '/'
    if args.temp_dir == 'AUTO/':
        args.temp_dir = tempfile.mkdtemp(prefix=args.data)
    if not args.temp_dir.endswith('/'):
        args.temp_dir += '/'
    if args.dbinfo_in == 'AUTO':
        args.dbinfo_in = args.data + 'db_info.txt'
    if args.db_dir == 'AUTO':
        args.db_dir = args.data + 'organism_files/'
    if args.input_type == 'AUTO':
        parts = args.reads.split('.')
        if parts[-1] == 'gz':
            parts = parts[:-1]
        if parts[-1] in ('fq', 'fastq'):
            args.input_type = 'fastq'
        elif parts[-1] in ('fa', 'fna', 'fasta'):
            args.input_type = 'fasta'
        else:
            sys.exit('Could not auto-determine file type. Use --input_type.')
    if args.sensitive and args.precise:
        sys.exit('You cannot use both --sensitive and --precise.')
    if args.sensitive:
        args.cutoff = 0.0
    elif args.precise:
        args.read_cutoff = 100
        args.min_abundance = 0.1
    args.db = args.temp_dir + 'cmashed_db.fna'
    args.dbinfo = args.temp_dir + 'subset_db_info.txt'
    args.dbinfo_out = args.dbinfo
    args.infiles = [args.reads]
    args.cmash_results = 'NONE'
    select.select_main(args)
    mapper.map_main(args)
    if not args.keep_temp_files:
        subprocess.Popen(['rm', '-r', args.temp_dir]).wait()


if __name__ == '__main__':
    main()
