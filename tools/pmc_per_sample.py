#!/usr/bin/env python3
"""pmc_per_sample.py <evidence dir> — wave instructions per step of 64 samples, from the pmc_summary.txt files of
profiles/collect_round.sh (headline in the directory itself, config_b / config_c / config_d / gated_98304 / fit5_81920 below it).
A step = one sample of one channel in each of a wave slot's 64 lanes; steps per decode = wave slots x frames x channels."""
import os, re, sys

def read(path):
    out, k = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.match(r"kernel (\S+)", line)
        if m:
            k = m.group(1); out[k] = {}; continue
        m = re.match(r"\s+(\S+)\s+n=\d+ mean=(\S+)", line)
        if m and k:
            out[k][m.group(1)] = float(m.group(2))
    return out

root = sys.argv[1]
print("# wave instructions per step of 64 samples (rocprofv3 --pmc, separate passes; pmc_summary.txt beside / below this file)")
for name, sub, packets, ch, fl in (("headline: 65 536 x 16-bit stereo", "", 65536, 2, 4096), ("config b: 4 096 x 16-bit stereo (16 packets per workgroup)", "config_b", 4096, 2, 4096),
                                   ("config c: 65 536 x 24-bit stereo, 1 shift byte", "config_c", 65536, 2, 4096),
                                   ("gated: 98 304 x 16-bit stereo", "gated_98304", 98304, 2, 4096),
                                   ("five workgroups per CU: 81 920 x 24-bit stereo", "fit5_81920", 81920, 2, 4096),
                                   ("config d: 16 384 x 24-bit 8-ch", "config_d", 16384, 8, 4096)):
    d = read(os.path.join(root, sub, "pmc_summary.txt"))
    lanes = 16 if packets == 4096 and ch == 2 else 64
    steps = packets / lanes * fl * ch
    for k, c in d.items():
        if c.get("SQ_INSTS_VALU", 0) < steps:  # kernels that did (next to) nothing for this configuration
            continue
        print("%s (%s), per wave step:" % (name, k))
        for key in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"):
            if key in c:
                print("  %-20s %8.2f" % (key, c[key] / steps))
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            print("  %-20s %8.3f" % ("WAIT_ANY/WAVE_CYCLES", c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
