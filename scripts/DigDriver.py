#!/usr/bin/env python
"""DigDriver.py -- command line of the burden tests on MI355X.

Drop-in for the reference's scripts/DigDriver.py: the same four sub-commands with the same positional
arguments and option names (DigDriver.py:160-275), and the same output, a tab-separated
``<outdir>/<outpfx>.results.txt`` with header and index column (DigDriver.py:38-43,115-118).  The statistics
run through libdig_hip.so; `model` may be the reference's HDF5 map (`*.h5`, read by io/h5lite.py +
io/pandas_fixed.py: no h5py or PyTables needed) or the directory mirror described in digdriver_amd/io/mapfile.py.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from digdriver_amd.driver_model import transfer_tools  # noqa: E402
from digdriver_amd.io import mapfile  # noqa: E402

PANELS = ['MSK_230', 'MSK_341', 'MSK_410', 'MSK_468', 'metabric_173', 'ucla_1202']
CGC_SETS = ['CGC_ALL', 'CGC_ONC', 'CGC_TSG']
SCALE_TYPES = ['genome', 'exome', 'sample', 'MSK_230', 'PCAWG_cds']


def _use_panel_dir(args):
    if getattr(args, 'panel_dir', None):
        transfer_tools.set_panel_dir(args.panel_dir)


def write_results(frame, args):
    os.makedirs(args.outdir, exist_ok=True)
    target = os.path.join(args.outdir, args.outpfx + '.results.txt')
    print('\tSaving results to {}'.format(target))
    mapfile.write_results_tsv(frame, target)               # the bytes of frame.to_csv(target, header=True, index=True, sep="\t")


def cmd_gene(args):
    _use_panel_dir(args)
    print('Running gene driver detection')
    res = transfer_tools.run_gene_model(
        args.fmut, args.model, scale_by_sample=args.scale_by_samples, pval_burden_nb=args.pval_burden,
        max_muts_per_sample=args.max_muts_per_sample, max_muts_per_gene_per_sample=args.max_muts_per_gene_per_sample,
        scale_factor=args.scale_factor_manual, scale_by_expectation=args.scale_by_expectation, cgc_genes=args.cgc_genes,
        fused=True)
    write_results(res, args)


def cmd_target(args):
    _use_panel_dir(args)
    print('Running MSK-IMPACT driver detection')
    res = transfer_tools.run_target_model(
        args.fmut, args.model, scale_by_sample=args.scale_by_samples, panel=args.panel,
        max_muts_per_sample=args.max_muts_per_sample, max_muts_per_gene_per_sample=args.max_muts_per_gene_per_sample,
        cgc_genes=args.cgc_genes, scale_factor=args.scale_factor_manual, drop_synonymous=False)
    write_results(res, args)


def _scale_mode(args):
    """Expectation scaling is the default; naming a scale type or a manual factor switches it off, and manual
    factors must come as a pair (DigDriver.py:74-80)."""
    args.scale_by_expectation = not (args.scale_type or args.scale_factor_manual)
    if args.scale_factor_manual or args.scale_factor_indel_manual:
        if not (args.scale_factor_manual and args.scale_factor_indel_manual):
            raise SystemExit("ERROR: must specify both --scale-factor-manual and --scale-factor-indel-manual.")


def cmd_element(args):
    _use_panel_dir(args)
    if not (args.f_bed or args.f_sites):
        raise SystemExit("ERROR: you must provide --f-bed or --f-sites.")
    print('Running user-defined element driver detection')
    _scale_mode(args)
    if args.f_sites:
        res = transfer_tools.run_sites_region_model(
            args.fmut, args.f_sites, args.model, args.pretrain_key, scale_factor=args.scale_factor_manual,
            scale_type=args.scale_type, scale_by_expectation=args.scale_by_expectation)
    else:
        res = transfer_tools.run_element_region_model(
            args.fmut, args.f_bed, args.model, args.pretrain_key, scale_type=args.scale_type,
            scale_factor=args.scale_factor_manual, scale_factor_indel=args.scale_factor_indel_manual,
            max_muts_per_sample=args.max_muts_per_sample, max_muts_per_elt_per_sample=args.max_muts_per_elt_per_sample,
            scale_by_expectation=args.scale_by_expectation, skip_pvals=args.skip_pvals, fused=True)
    for col in ('OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL'):       # integer columns in the TSV (DigDriver.py:108-112)
        if col in res.columns:
            res[col] = res[col].astype(int)
    write_results(res, args)


def cmd_quick(args):
    _use_panel_dir(args)
    if not (args.f_elts_bed or args.region_str):
        raise SystemExit("ERROR: you must provide --f_elts_bed or --region_str.")
    from digdriver_amd.driver_model import onthefly_tools
    print('Running user-defined element driver detection')
    _scale_mode(args)
    res = onthefly_tools.DIG_onthefly(
        args.model, args.fmut, args.f_fasta, f_elts_bed=args.f_elts_bed, region_str=args.region_str,
        scale_factor=args.scale_factor_manual, scale_factor_indel=args.scale_factor_indel_manual,
        scale_type=args.scale_type, max_muts_per_sample=args.max_muts_per_sample,
        max_muts_per_elt_per_sample=args.max_muts_per_elt_per_sample, scale_by_expectation=args.scale_by_expectation,
        skip_pvals=args.skip_pvals)
    for col in ('OBS_SAMPLES', 'OBS_SNV', 'OBS_INDEL'):
        if col in res.columns:
            res[col] = res[col].astype(int)
    write_results(res, args)


def _common(p, element_caps):
    p.add_argument('fmut', type=str, help='annotated mutation file (DigPreprocess.py annotMutationFile format)')
    p.add_argument('model', type=str, help='pretrained mutation map')
    # not a reference option: the reference finds its gene panels (genes_CGC_ALL.txt, genes_MSK_341.txt, ...) inside its
    # installed package; here they are looked up in this directory, then $DIG_DATA_DIR, digdriver_amd/data/ and an
    # installed DIGDriver package
    p.add_argument('--panel-dir', type=str, default=None, help='directory holding the gene panel files genes_<NAME>.txt')
    return p


def _output(p):
    p.add_argument('--outpfx', type=str, required=True, help='prefix of the results file')
    p.add_argument('--outdir', type=str, required=True, help='directory for the results file')


def parse_args(text=None):
    parser = argparse.ArgumentParser(description='Burden tests for cancer driver elements (MI355X build).')
    sub = parser.add_subparsers()

    g = _common(sub.add_parser('geneDriver', help='test every gene of a cohort'), False)
    _output(g)
    g.add_argument('--max-muts-per-sample', type=int, default=3e9, help='drop samples with more mutations than this')
    g.add_argument('--max-muts-per-gene-per-sample', type=int, default=3e9, help='cap of mutations one sample adds to a gene')
    g.add_argument('--scale-by-mutations', action='store_false', default=True, dest="scale_by_expectation",
                   help='scale by mutation counts instead of expected synonymous mutations')
    g.add_argument('--scale-by-samples', action='store_true', default=False, help='scale by the number of samples')
    g.add_argument('--scale-factor-manual', default=None, type=float, help='use this scale factor')
    g.add_argument('--cgc-genes', choices=CGC_SETS, default=False, help='restrict to a Cancer Gene Census set')
    g.add_argument('--no-pval-burden', dest='pval_burden', action='store_false', default=True,
                   help='skip the burden p-values')
    g.set_defaults(func=cmd_gene)

    t = _common(sub.add_parser('targetDriver', help='test the genes of a targeted sequencing panel'), False)
    _output(t)
    t.add_argument('--panel', type=str, choices=PANELS, help='gene panel')
    t.add_argument('--max-muts-per-sample', type=int, default=3e9, help='drop samples with more mutations than this')
    t.add_argument('--max-muts-per-gene-per-sample', type=int, default=3e9, help='cap of mutations one sample adds to a gene')
    t.add_argument('--scale-by-samples', action='store_true', default=False, help='scale by the number of samples')
    t.add_argument('--scale-factor-manual', default=None, type=float, help='use this scale factor')
    t.add_argument('--cgc-genes', choices=CGC_SETS, default=False, help='restrict to a Cancer Gene Census set')
    t.set_defaults(func=cmd_target)

    for name, func, extra in (('elementDriver', cmd_element, 'element'), ('quickDriver', cmd_quick, 'quick')):
        e = _common(sub.add_parser(name, help='test user-defined elements' if extra == 'element'
                                   else 'test ad-hoc elements or a region string (no pretrained element model)'), True)
        if extra == 'element':
            e.add_argument('pretrain_key', type=str, help='key of the pretrained element model inside the map')
            e.add_argument('--f-bed', type=str, default="", help='bed12 file the element model was pretrained on')
            e.add_argument('--f-sites', type=str, default="", help='sites file the element model was pretrained on (SNVs only)')
        else:
            e.add_argument('f_fasta', type=str, help='reference genome FASTA (hg19)')
            e.add_argument('--f_elts_bed', type=str, default="", help='bed12 file of elements')
            e.add_argument('--region_str', type=str, default="", help='region as chr{}:start-end')
        _output(e)
        e.add_argument('--max-muts-per-sample', type=int, default=3e9, help='drop samples with more mutations in elements than this')
        e.add_argument('--max-muts-per-elt-per-sample', type=int, default=3e9, help='cap of mutations one sample adds to an element')
        e.add_argument('--scale-type', default=None, choices=SCALE_TYPES, help='how to derive the cohort scale factor')
        e.add_argument('--scale-factor-manual', default=None, type=float, help='use this SNV scale factor')
        e.add_argument('--skip_pvals', default=False, action='store_true', help='expected counts only')
        e.add_argument('--scale-factor-indel-manual', default=None, type=float, help='use this indel scale factor')
        e.set_defaults(func=func)

    return parser.parse_args(text.split()) if text else parser.parse_args()


def main(text=None):
    cli = parse_args(text)
    if cli.func in (cmd_gene, cmd_target, cmd_element):
        # these three read frames, hand numpy arrays to the library's `_host` entry points and write a frame: no tensor is ever made,
        # so PyTorch (1.5 s of import and device-layer start for milliseconds of GPU work) is not loaded at all
        from digdriver_amd import _lib
        _lib.TORCH_FREE = True
        _lib.prewarm_in_background()            # (the HIP runtime starts while pandas is imported and the files are parsed)
    cli.func(cli)
    if os.environ.get("DIG_CLI_ASSERT_NO_TORCH") == "1" and cli.func in (cmd_gene, cmd_target, cmd_element):
        assert "torch" not in sys.modules, "a torch-free sub-command imported torch"         # (tests: the claim above)


if __name__ == "__main__":
    main()
