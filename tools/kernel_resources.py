#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel in a gfx950 assembly file (hipcc --save-temps=obj ... ; the
amdhsa.kernels metadata at its end).  Usage: kernel_resources.py FILE.s [name filter]"""
import re
import sys


def main(path, flt=""):
    text = open(path).read()
    meta = text[text.index("amdhsa.kernels:"):]
    print("%-64s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "sgpr", "v_spill", "s_spill", "scratch", "lds"))
    for blk in meta.split("  - .agpr_count:")[1:]:
        get = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, "?"])[1]
        name = get("name")
        short = re.sub(r"^_ZN\d+toast_hip\d+fused_fft\d+|^_ZN\d+_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"EvNS0_6ParamsE$|Ev.*$", "", short)
        if flt and flt not in name:
            continue
        print("%-64s %5s %5s %7s %7s %8s %7s" % (short[:64], get("vgpr_count"), get("sgpr_count"), get("vgpr_spill_count"),
                                                 get("sgpr_spill_count"), get("private_segment_fixed_size"),
                                                 get("group_segment_fixed_size")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
