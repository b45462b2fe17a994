#!/usr/bin/env python
"""Share of a host file's non-blank lines that occur verbatim (whitespace-normalised) in the same-named reference file.
Developer tool for keeping the host mirror a re-design rather than a transcription.  python tools/linematch.py [files]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
PAIRS = {
    "digdriver_amd/driver_model/transfer_tools.py": "DIGDriver/driver_model/transfer_tools.py",
    "digdriver_amd/driver_model/onthefly_tools.py": "DIGDriver/driver_model/onthefly_tools.py",
    "digdriver_amd/data_tools/mutation_tools.py": "DIGDriver/data_tools/mutation_tools.py",
    "digdriver_amd/region_model/region_model_tools.py": "DIGDriver/region_model/region_model_tools.py",
    "digdriver_amd/sequence_model/sequence_tools.py": "DIGDriver/sequence_model/sequence_tools.py",
    "digdriver_amd/sequence_model/genic_driver_tools.py": "DIGDriver/sequence_model/genic_driver_tools.py",
    "digdriver_amd/sequence_model/nb_model.py": "DIGDriver/sequence_model/nb_model.py",
    "scripts/DigDriver.py": "scripts/DigDriver.py",
    "scripts/DigPretrain.py": "scripts/DigPretrain.py",
}


def norm(line):
    return re.sub(r"\s+", " ", line.strip())


def main():
    files = sys.argv[1:] or list(PAIRS)
    for f in files:
        ours = [norm(l) for l in open(os.path.join(ROOT, f)) if l.strip()]
        ref = {norm(l) for l in open(os.path.join(REF, PAIRS[f])) if l.strip()}
        hit = [l for l in ours if l in ref]
        long_hit = [l for l in hit if len(l) > 25]
        print("%-55s %4d / %4d = %4.1f %%   (longer than 25 chars: %d)" % (f, len(hit), len(ours), 100.0 * len(hit) / max(len(ours), 1), len(long_hit)))
        if "-v" in os.environ.get("LM", ""):
            for l in long_hit:
                print("      ", l)


if __name__ == "__main__":
    main()
