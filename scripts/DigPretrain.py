#!/usr/bin/env python
"""DigPretrain.py -- assemble a pretrained mutation map on MI355X.

Drop-in for the sub-commands of the reference's scripts/DigPretrain.py (:280-477) that lie on the hot path:

    regionModel     k-fold CNN+GP results -> idx, mappability, region_params (+ mutation counts)   (:31-100)
    countMutations  cohort-level mutation counts stored as attributes                               (:102-177)
    sequenceModel   sequence_model_192 / sequence_model_64                                          (:179-208)
    genicModel      genic_model frame                                                               (:226-237)
    elementModel    <save_key> element frame                                                        (:239-268)
    tiledModel      <save_key> tile frame                                                           (:271-278)

Same positional arguments and option names.  `countNonc_context` is the reference's own deprecated
sub-command (it calls a function that does not exist, DigPretrain.py:222) and is not provided.
Maps may be HDF5 (`*.h5`, read and written by io/h5lite.py + io/pandas_fixed.py: no h5py or PyTables needed) or the directory mirror (digdriver_amd/io/mapfile.py).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from digdriver_amd.data_tools import mutation_tools                      # noqa: E402
from digdriver_amd.driver_model import transfer_tools                    # noqa: E402
from digdriver_amd.io import mapfile                                     # noqa: E402
from digdriver_amd.region_model import region_model_tools               # noqa: E402
from digdriver_amd.sequence_model import genic_driver_tools, sequence_tools   # noqa: E402


def get_cpus():
    """auxilaries/utils.py:3-8 (only used as the default of --n-procs, which the GPU path ignores)."""
    try:
        return min(max(1, (os.cpu_count() or 3) - 2), 20)
    except Exception:
        return 5


def pretrain_region_model(args):
    if not os.path.isdir(args.kfold_dir):
        raise SystemExit("The supplied kfold results path {} is not a directory.".format(args.kfold_dir))
    if not args.cohort_name:
        args.cohort_name = os.path.basename(os.path.normpath(args.kfold_dir))
    idx_all = mapfile.read_array(args.train_h5, 'idx')
    mapp = mapfile.read_array(args.train_h5, 'mappability')
    df = region_model_tools.kfold_results(args.kfold_dir, args.cohort_name, key=args.key)
    if os.path.exists(args.outputFile) and not args.append:
        raise SystemExit("{} exists; pass --append to add to it.".format(args.outputFile))
    with mapfile.batch(args.outputFile):
        mapfile.write_array(args.outputFile, 'idx', idx_all.astype(np.int32))
        mapfile.write_array(args.outputFile, 'mappability', mapp.astype(np.float32))
        mapfile.write_attrs(args.outputFile, cohort_name=args.cohort_name, mappability_threshold=args.map_thresh)
        mapfile.write_frame(args.outputFile, 'region_params', df)
    if args.fmut:
        print('Adding mutation counts...')
        count_training_mutations(args)


def count_training_mutations(args):
    df = mapfile.read_frame(args.outputFile, 'region_params')
    flag = df.FLAG.values.astype(bool)
    attrs = {'N_MUT_TOTAL': int(df.Y_TRUE.sum()), 'N_MUT_TRAIN': int(df.Y_TRUE.values[~flag].sum())}
    df_mut = mutation_tools.read_mutation_file(args.fmut, drop_duplicates=True)
    cds = df_mut[df_mut.ANNOT != 'Noncoding']
    attrs['N_SAMPLES'] = int(len(df_mut.SAMPLE.unique()))
    attrs['N_MUT_CDS'] = int(len(cds))
    attrs['N_MUT_SAMPLE_CDS'] = int(len(cds))      # the reference stores N_MUT_CDS under this name too (DigPretrain.py:161)
    nonsyn = cds[(cds.ANNOT != 'Synonymous') & (cds.ANNOT != 'Essential_Splice') & (cds.ANNOT != 'Noncoding')]
    for attr, panel in (('MSK_230', 'MSK_230'), ('MSK_341', 'MSK_341'), ('MSK_410', 'MSK_410'), ('MSK_468', 'MSK_468'),
                        ('metabric_173', 'metabric_173'), ('ucla_1202', 'ucla_1202')):
        try:
            genes = transfer_tools._read_gene_panel(panel)
        except FileNotFoundError:
            continue                               # panel list not installed
        sub = nonsyn[nonsyn.GENE.isin(genes)]
        attrs['N_MUT_' + attr] = int(len(sub))
        attrs['N_MUT_SAMPLE_' + attr] = int(sub.groupby(['GENE', 'SAMPLE']).ngroups)
        if panel == 'MSK_230':
            attrs['N_SAMPLE_MSK_230'] = int(len(sub.SAMPLE.unique()))
    mapfile.write_attrs(args.outputFile, **attrs)


def pretrain_sequence_model(args):
    print('Loading genome-wide context counts')
    df_genome = mapfile.read_frame(args.genome_counts, 'all_window_genome_counts')
    idx = mapfile.read_array(args.genome_counts, 'idx')
    mapp = mapfile.read_array(args.genome_counts, 'mappability')
    keep = mapp > args.map_thresh
    S_genome = df_genome[keep].sum(axis=0)
    print('Loading mutation file')
    df_mut = mutation_tools.read_mutation_file(args.fmut, drop_duplicates=True)
    df_mut = df_mut[df_mut.ANNOT != 'INDEL']
    print('Training sequence model')
    f192, f64 = sequence_tools.train_sequence_model(idx[keep], df_mut, S_genome)
    print('Saving sequence models to {}'.format(args.output_h5))
    with mapfile.batch(args.output_h5):
        mapfile.write_frame(args.output_h5, 'sequence_model_192', f192)
        mapfile.write_frame(args.output_h5, 'sequence_model_64', f64)


def pretrain_genic_model(args):
    print('Running Genic model')
    frame = genic_driver_tools.genic_model_parallel(args.f_pretrained, args.f_genic, args.N_procs,
                                                    counts_key=args.counts_key, indels_direct=args.indels_direct)
    mapfile.write_frame(args.output_h5 or args.f_pretrained, 'genic_model', frame)


def pretrain_nonc_model(args):
    print('Pretraining element model')
    frame = genic_driver_tools.nonc_model_parallel(args.f_pretrained, args.f_element_data, args.save_key, args.N_procs,
                                                   indels_direct=args.indels_direct)
    print("saving")
    mapfile.write_frame(args.output_h5 or args.f_pretrained, args.save_key, frame)


def pretrain_tiled(args):
    frame = genic_driver_tools.tiled_model_parallel(args.f_pretrained, args.f_element_data, args.save_key, args.N_procs)
    print("saving")
    mapfile.write_frame(args.output_h5 or args.f_pretrained, args.save_key, frame)


def parse_args(text=None):
    parser = argparse.ArgumentParser(description='Build a pretrained DIG mutation map (MI355X build).')
    sub = parser.add_subparsers()

    a = sub.add_parser('regionModel', help='region parameters from a finished CNN+GP k-fold run')
    a.add_argument('kfold_dir', type=str, help='directory with the k-fold result files')
    a.add_argument('train_h5', type=str, help='training data container (idx, mappability)')
    a.add_argument('outputFile', help='mutation map to write')
    a.add_argument('--cohort-name', type=str, default='', help='cohort key inside the k-fold results')
    a.add_argument('--key', type=str, default='held-out', help='result group to load')
    a.add_argument('--map-thresh', type=float, default=0.5, help='mappability threshold used in training')
    a.add_argument('--mutation-file', type=str, default=None, dest='fmut', help='mutation file for the cohort counts')
    a.add_argument('--cds-file', type=str, default="../data/dndscv_gene_cds.bed.gz", help='CDS bed file (unused)')
    a.add_argument('--append', action='store_true', default=False, help='add to an existing map')
    a.set_defaults(func=pretrain_region_model)

    a1 = sub.add_parser('countMutations', help='store cohort mutation counts in a map')
    a1.add_argument('--outputFile', required=True, help='mutation map')
    a1.add_argument('--mutation-file', required=True, type=str, dest='fmut', help='mutation file')
    a1.set_defaults(func=count_training_mutations)

    b = sub.add_parser('sequenceModel', help='trinucleotide sequence model from genome counts + mutations')
    b.add_argument('fmut', help='annotated mutation file')
    b.add_argument('genome_counts', help='genome-wide context counts container')
    b.add_argument('output_h5', help='mutation map to write into')
    b.add_argument('--map-thresh', default=0.5, type=float, help='minimum bin mappability')
    b.set_defaults(func=pretrain_sequence_model)

    d = sub.add_parser('genicModel', help='per-gene parameters')
    d.add_argument('f_pretrained', help='map with region and sequence models')
    d.add_argument('f_genic', help='preprocessed gene data container')
    d.add_argument('--counts-key', default="window_10kb/counts", help='window counts key in f_genic')
    d.add_argument('--output_h5', help='write here instead of f_pretrained')
    d.add_argument('--indels-direct', action='store_true', default=False, help='use a separate indel region model')
    d.add_argument('--n-procs', default=get_cpus(), type=int, dest='N_procs', help='accepted for compatibility')
    d.set_defaults(func=pretrain_genic_model)

    for name, func in (('elementModel', pretrain_nonc_model), ('tiledModel', pretrain_tiled)):
        e = sub.add_parser(name, help='per-element parameters' if name == 'elementModel' else 'per-tile parameters')
        e.add_argument('f_pretrained', help='map with region and sequence models')
        e.add_argument('f_element_data', help='precounted element / region contexts')
        e.add_argument('save_key', help='key of the element set and of the frame to write')
        e.add_argument('--output_h5', help='write here instead of f_pretrained')
        if name == 'elementModel':
            e.add_argument('--indels-direct', action='store_true', default=False, help='use a separate indel region model')
        e.add_argument('--n-procs', default=get_cpus(), type=int, dest='N_procs', help='accepted for compatibility')
        e.set_defaults(func=func)

    return parser.parse_args(text.split()) if text else parser.parse_args()


def main(text=None):
    cli = parse_args(text)
    if cli.func is pretrain_nonc_model:
        # element frames are made from numpy arrays by the library's `_host` entry points: PyTorch (1.5 s of import and device-layer
        # start) is not loaded for this sub-command (scripts/DigDriver.py does the same for its single-cohort commands)
        from digdriver_amd import _lib
        _lib.TORCH_FREE = True
        _lib.prewarm_in_background()            # (the HIP runtime starts while pandas is imported and the files are parsed)
    cli.func(cli)
    if os.environ.get("DIG_CLI_ASSERT_NO_TORCH") == "1" and cli.func is pretrain_nonc_model:
        assert "torch" not in sys.modules, "a torch-free sub-command imported torch"


if __name__ == "__main__":
    main()
