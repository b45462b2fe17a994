#!/usr/bin/env python
"""Developer tool: static instruction mix of the statistics kernel's tile loop (the body between the loop header and the
last streaming store of element_stats_stream_fused_kernel<1024, true, 0>), from hipcc -S.  A proxy for the dynamic count
(rocprofv3 SQ_INSTS_VALU: 807 per tile in round 3) that needs no GPU.   python tools/asm_loop_stats.py [extra hipcc flags]"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/dis/dig_nb_stats.s"
os.makedirs("/tmp/dis", exist_ok=True)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + ROOT + "/include", "-mllvm",
       "-disable-machine-licm", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-S", "--cuda-device-only", "-o", out,
       ROOT + "/digdriver_amd/csrc/dig_nb.hip"] + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3dig33element_stats_stream_fused_kernelILi1024ELb1ELi3ELb1E"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
k = lines[start:end]
stores = [i for i, l in enumerate(k) if "global_store" in l and " nt" in l]
first = stores[0]
hdr = max(i for i in range(first) if "Loop Header: Depth=1" in k[i])
last = max(i for i in stores if i - first < 80)
body = [l.strip() for l in k[hdr:last + 1] if re.match(r"^\s+[a-z]", l)]
ops = collections.Counter(l.split()[0] for l in body)
def tot(pred): return sum(n for o, n in ops.items() if pred(o))
valu = tot(lambda o: o.startswith("v_"))
print("loop body: %d instructions, %d VALU, %d SALU, %d vector memory, %d LDS, %d branches" % (
    len(body), valu, tot(lambda o: o.startswith("s_") and not o.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop"))),
    tot(lambda o: o.startswith("global_")), tot(lambda o: o.startswith("ds_")), tot(lambda o: o.startswith(("s_cbranch", "s_branch")))))
groups = [("f64 fma/fmac", ("v_fma_f64", "v_fmac_f64")), ("f64 add", ("v_add_f64",)), ("f64 mul", ("v_mul_f64",)),
          ("moves", ("v_mov_b32", "v_mov_b64", "v_accvgpr")), ("selects", ("v_cndmask",)), ("compares", ("v_cmp",)),
          ("readlane/readfirstlane", ("v_readlane", "v_readfirstlane", "v_writelane")), ("converts/ldexp/rcp/div", ("v_cvt", "v_ldexp", "v_rcp", "v_rsq", "v_div", "v_rndne", "v_max_f64", "v_min_f64")),
          ("64-bit address", ("v_lshl_add_u64", "v_lshlrev_b64", "v_mad_u64"))]
seen = 0
for name, pre in groups:
    n = tot(lambda o: o.startswith(pre))
    seen += n
    print("  %-28s %4d" % (name, n))
print("  %-28s %4d" % ("other VALU (integer ...)", valu - seen))
m = re.search(r"\.amdhsa_next_free_vgpr (\d+)", "\n".join(lines[end:end + 200]))
sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", "\n".join(lines[end:end + 200]))
print("VGPRs %s SGPRs %s" % (m and m.group(1), sg and sg.group(1)))
