"""Command-line surface shared by the three entry points.

The flag names, defaults, types, choices and sentinels ('AUTO', 'NONE', 'AUTO/') are the reference's
(/root/reference/scripts/metalign.py:8-40, select_db.py:5-24, map_and_profile.py:20-45) so that existing
invocations keep working; tests/test_select_and_cli.py checks them against a fixture dumped from the reference's
own parsers.  They are declared once here, as data, and each tool picks the subset it accepts.
"""
import argparse
import sys

# name -> (argparse kwargs).  Store-true switches carry action='store_true'.
_OPTIONS = {
    'cmash_results': dict(default='NONE', help='Existing containment CSV to reuse (skips the GPU pre-filter).'),
    'cutoff': dict(type=float, default=0.01, help='Keep organisms whose containment index is at least this (default 0.01).'),
    'db': None,  # differs per tool, see below
    'db_dir': dict(default='AUTO', help='Directory holding one FASTA(.gz) per organism of the full database.'),
    'dbinfo': dict(default='AUTO', help='db_info file to profile against (default: data/db_info.txt).'),
    'dbinfo_in': dict(default='AUTO', help='db_info of the full database (default: data/db_info.txt).'),
    'dbinfo_out': dict(default='AUTO', help='Where the subset db_info goes (default: temp_dir/subset_db_info.txt).'),
    'input_type': None,  # differs per tool
    'keep_temp_files': dict(action='store_true', help='Leave the temporary directory in place.'),
    'length_normalize': dict(action='store_true', help='Divide base counts by genome length.'),
    'low_mem': dict(action='store_true', help='Low-memory handling of multimapped reads (inexact).'),
    'min_abundance': dict(type=float, default=10**-4, help='Do not report taxa below this abundance (default 1e-4).'),
    'no_quantify_unmapped': dict(action='store_true', help='Ignore unmapped reads when computing abundances.'),
    'output': dict(default='abundances.tsv', help='CAMI profile to write (default abundances.tsv).'),
    'pct_id': dict(type=float, default=0.5, help='Minimum matched fraction of an alignment for it to count (default 0.5).'),
    'precise': dict(action='store_true', help='Precise mode: read_cutoff 100, min_abundance 0.1.'),
    'rank_renormalize': dict(action='store_true', help='Rescale every rank to the mapped percentage.'),
    'read_cutoff': dict(type=int, default=1, help='An organism needs MORE than this many unique reads (default 1).'),
    'sampleID': dict(default='NONE', help='Sample ID written to the profile header (default: input file names).'),
    'sensitive': dict(action='store_true', help='Sensitive mode: containment cutoff 0.'),
    'strain_level': dict(action='store_true', help='Keep every strain above the cutoff instead of one per species.'),
    'temp_dir': dict(default='AUTO/', help='Directory for intermediate files (default: a fresh one under data/).'),
    'threads': dict(type=int, default=4, help='Threads handed to the aligner (default 4).'),
    'verbose': dict(action='store_true', help='Progress messages.'),
    # build-only additions; defaults reproduce the reference behaviour
    'sketch_table': dict(default='AUTO', help='Genome sketch table directory (default: data/sketch_table).'),
    'min_count': dict(type=int, default=2, help='A read k-mer must occur this often to count (kmc -ci, default 2).'),
    'sketch_size': dict(type=int, default=0, help='Read sketch size per k; 0 keeps every hash up to the table maximum.'),
    'kmer_match': dict(default='identity', choices=['identity', 'identity_only', 'hash'],
                       help='A reference-pipeline sketch table: a read k-mer meets a sketched one by what it IS, as kmc and kmc_tools '
                            'intersect compare k-mers (default where it is the faster way: a table that stores its k-mers, the largest k '
                            'from 25 to 64; "hash" otherwise), wherever that can run at all (identity_only: the largest k from 15 to 64), or '
                            'by its MurmurHash3 value.'),
    'device_multimap': dict(action='store_true', help='Resolve multimapped reads on the GPU (their lists never leave '
                            'the device; abundances equal the default path to ~1e-15 relative, not byte for byte).'),
}

_READ_TYPES = ['fastq', 'fasta', 'AUTO']

_TOOLS = {
    'metalign': dict(
        description='Runs full metalign pipeline on input reads file(s).',
        positionals=[('reads', dict(help='Reads file (FASTA / FASTQ, optionally .gz).')),
                     ('data', dict(help='data/ directory (db_info.txt, organism_files/, sketch_table/).'))],
        options=['cutoff', 'db_dir', 'dbinfo_in', 'keep_temp_files', ('input_type', _READ_TYPES), 'length_normalize',
                 'low_mem', 'min_abundance', 'no_quantify_unmapped', 'output', 'pct_id', 'precise', 'rank_renormalize',
                 'read_cutoff', 'sampleID', 'sensitive', 'strain_level', 'temp_dir', 'threads', 'verbose',
                 'sketch_table', 'min_count', 'sketch_size', 'kmer_match', 'device_multimap']),
    'select_db': dict(
        description='Run CMash and select a subset of the whole database to align to.',
        positionals=[('reads', dict(help='Reads file (FASTA / FASTQ, optionally .gz).')),
                     ('data', dict(help='data/ directory (db_info.txt, organism_files/, sketch_table/).'))],
        options=['cmash_results', 'cutoff', ('db', 'AUTO', 'Subset database FASTA to write (default: temp_dir/cmashed_db.fna).'),
                 'db_dir', 'dbinfo_in', 'dbinfo_out', ('input_type', _READ_TYPES), 'keep_temp_files', 'strain_level',
                 'temp_dir', 'threads', 'sketch_table', 'min_count', 'sketch_size', 'kmer_match']),
    'map_and_profile': dict(
        description='Compute abundance estimations for species in a sample.',
        positionals=[('infiles', dict(nargs='+', help='SAM file(s), or reads file(s) to align with minimap2.')),
                     ('data', dict(help='data/ directory.'))],
        options=[('db', 'NONE', 'Database FASTA from select_db (needed unless the inputs are SAM files).'), 'dbinfo',
                 ('input_type', ['fastq', 'fasta', 'sam', 'AUTO']), 'length_normalize', 'low_mem', 'min_abundance',
                 'rank_renormalize', 'output', 'pct_id', 'no_quantify_unmapped', 'read_cutoff', 'sampleID', 'threads',
                 'verbose', 'device_multimap']),
}


def parser_for(tool):
    spec = _TOOLS[tool]
    p = argparse.ArgumentParser(description=spec['description'])
    for name, kw in spec['positionals']:
        p.add_argument(name, **kw)
    for opt in spec['options']:
        if isinstance(opt, tuple) and opt[0] == 'input_type':
            p.add_argument('--input_type', default='AUTO', choices=opt[1],
                           help='Input format; AUTO looks at the file extension.')
        elif isinstance(opt, tuple) and opt[0] == 'db':
            p.add_argument('--db', default=opt[1], help=opt[2])
        else:
            p.add_argument('--' + opt, **_OPTIONS[opt])
    return p


def with_slash(path):
    return path if path.endswith('/') else path + '/'


_EXT = {'fq': 'fastq', 'fastq': 'fastq', 'fa': 'fasta', 'fna': 'fasta', 'fasta': 'fasta'}


def sniff_reads_type(path):
    """'fastq' / 'fasta' from the extension (a trailing .gz is ignored); exits like the reference when unknown."""
    parts = path.split('.')
    ext = parts[-2] if parts[-1] == 'gz' and len(parts) > 1 else parts[-1]
    kind = _EXT.get(ext)
    if kind is None:
        sys.exit('Could not auto-determine file type. Use --input_type.')
    return kind
