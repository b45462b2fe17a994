#!/usr/bin/env python
"""DigPreprocess.py -- sequence-context preprocessing on MI355X.

The sub-commands of the reference's scripts/DigPreprocess.py that feed the burden-test path with context counts
(the annotation sub-commands need bedtools / R and are out of scope):

    countGenomeContext        window context counts of a genome          (DigPreprocess.py:19-73)
    initialize_f_data         start an element-data container            (:147-153)
    preprocess_element_model  per-element L counts from bed12 + FASTA    (:129-145)
    preprocess_tiled          L counts of a tiled genome                 (:155-164)

Same positional arguments and option names.  Sequence is read once into a 4-bit packed array (cached next to the
FASTA) and counted by dig_count_contexts instead of per-region pysam fetches.
"""
import argparse
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from digdriver_amd.io import mapfile                                       # noqa: E402
from digdriver_amd.sequence_model import sequence_tools                    # noqa: E402


def count_genome_context(args):
    if bool(args.h5) == bool(args.bed):
        raise SystemExit("Exactly one of --h5 or --bed must be supplied.")
    if (args.up, args.down) != (1, 1):
        raise SystemExit("This build counts trinucleotide contexts (--up 1 --down 1).")
    if args.map_file:
        raise SystemExit("--map-file needs the bigWig reader of the reference's preprocessing stack (out of scope).")
    if args.h5:
        df_bed = pd.DataFrame(mapfile.read_array(args.h5, 'idx'))
    else:
        df_bed = pd.read_table(args.bed, header=None, low_memory=False)
        df_bed[0] = df_bed[0].astype(str)
        df_bed = df_bed[df_bed[0].isin([str(i) for i in range(1, 23)])].copy()      # autosomes (:37-39)
        df_bed[0] = df_bed[0].astype(int)
    df_bed = df_bed.sort_values(by=[0, 1])
    print('Counting nucleotide contexts in {} regions'.format(len(df_bed)))
    df = sequence_tools.count_contexts_in_bed(args.fasta, df_bed, n_up=1, n_down=1)
    idx = df_bed.iloc[:, 0:3].values
    print('Saving context counts to {}'.format(args.fout))
    with mapfile.batch(args.fout):
        mapfile.write_frame(args.fout, 'genome_counts', df.sum(axis=0).to_frame('COUNT'))
        mapfile.write_frame(args.fout, 'all_window_genome_counts', df)
        mapfile.write_array(args.fout, 'idx', idx.astype(np.int32))
        mapfile.write_attrs(args.fout, n_up=1, n_down=1, collapse=0)


def initialize_data(args):
    idx = mapfile.read_array(args.f_genome_counts, 'idx')
    if not mapfile.has_key(args.f_genome_counts, 'all_window_genome_counts'):
        raise SystemExit("f_genome_counts does not hold 'all_window_genome_counts'.")
    sequence_tools.initialize_nonc_data(args.f_annot_data, args.f_genome_counts, int(idx[0, 2] - idx[0, 1]))


def preprocess_nonc_contexts(args):
    if args.f_sites:
        print("preprocessing sites data")
        sequence_tools.preprocess_sites(args.f_sites, args.f_element_data, args.f_pretrained, args.save_key, args.window)
        return
    if not args.f_element_bed:
        raise SystemExit("ERROR: need to pass in an elements file (--f-bed) for preprocessing")
    print("Preprocessing elements")
    L = sequence_tools.precount_region_contexts_parallel(args.f_element_bed, args.f_fasta, args.N_procs, args.window,
                                                         args.use_sub_elts)
    print('window counts by elt')
    sequence_tools.preprocess_nonc(args.f_element_bed, args.f_element_data, args.f_pretrained, L, args.save_key, args.window)


def preprocess_tiled(args):
    print("Counting sequence contexts in regions")
    L = sequence_tools.precount_region_contexts_parallel(args.f_nonc_bed, args.f_fasta, args.N_procs, args.window, False)
    mapfile.write_frame(args.f_nonc_data, "{}/L_counts".format(args.save_key), L.astype(np.int32))


def parse_args(text=None):
    parser = argparse.ArgumentParser(description='Sequence-context preprocessing for the burden-test path (MI355X build).')
    sub = parser.add_subparsers()
    a = sub.add_parser('countGenomeContext', help='trinucleotide context counts of genome windows')
    a.add_argument('fasta', type=str, help='reference genome FASTA')
    a.add_argument('fout', type=str, help='container to write')
    a.add_argument('--h5', type=str, default='', help='container holding the windows as `idx`')
    a.add_argument('--bed', type=str, default='', help='headerless bed file of windows')
    a.add_argument('--up', type=int, default=1, help='bases upstream (1)')
    a.add_argument('--down', type=int, default=1, help='bases downstream (1)')
    a.add_argument('--n-procs', type=int, default=1, help='accepted for compatibility')
    a.add_argument('--map-file', type=str, default='', help='not supported here')
    a.add_argument('--map-thresh', type=float, default=0.5, help='unused')
    a.set_defaults(func=count_genome_context)

    e = sub.add_parser('preprocess_element_model', help='per-element context counts from a bed12 file')
    e.add_argument('f_element_data', help='element-data container (see initialize_f_data)')
    e.add_argument('f_pretrained', help='any pretrained map (kept for compatibility)')
    e.add_argument('f_fasta', help='reference genome FASTA (hg19)')
    e.add_argument('save_key', help='key of the element set')
    e.add_argument('--f-bed', dest='f_element_bed', help='bed12 file of the elements')
    e.add_argument('--f-sites', type=str, default=None, help='sites file (element name in the SAMPLE column)')
    e.add_argument('--ignore-sub_elts', action='store_false', default=True, dest='use_sub_elts',
                   help='count whole element spans instead of blocks')
    e.add_argument('--n-procs', default=1, type=int, dest='N_procs', help='accepted for compatibility')
    e.add_argument('--window', type=int, default=10000, help='window size in bp')
    e.set_defaults(func=preprocess_nonc_contexts)

    f = sub.add_parser('initialize_f_data', help='start an element-data container from genome window counts')
    f.add_argument('f_annot_data', help='container to create')
    f.add_argument('f_genome_counts', help='output of countGenomeContext')
    f.set_defaults(func=initialize_data)

    g = sub.add_parser('preprocess_tiled', help='context counts of a tiled genome')
    g.add_argument('f_nonc_bed', help='bed file of the tiles')
    g.add_argument('f_nonc_data', help='element-data container')
    g.add_argument('f_fasta', help='reference genome FASTA')
    g.add_argument('--n-procs', default=1, type=int, dest='N_procs', help='accepted for compatibility')
    g.add_argument('window', type=int, default=10000, help='window size in bp')
    g.add_argument('save_key', help='key of the tile set')
    g.set_defaults(func=preprocess_tiled)
    return parser.parse_args(text.split()) if text else parser.parse_args()


if __name__ == "__main__":
    cli = parse_args()
    cli.func(cli)
