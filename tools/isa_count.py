#!/usr/bin/env python3
"""Instruction accounting of one kernel from hipcc's --save-temps assembly (gfx950).

  hipcc ... --save-temps -c csrc/grt_api.hip        # leaves grt_api-hip-amdgcn-amd-amdhsa-gfx950.s
  tools/isa_count.py FILE.s KERNEL_SUBSTRING [--blocks 15,17,18,19:6,21:2,50] [--list]

Splits the kernel body into basic blocks (labels .LBBn_m and the fall-through '; %bb.m' markers), classifies every
instruction and prints the counts per block.  --blocks names the blocks one trip of the hot loop executes (m or m:times;
'bb.18' style ids are given as b18) and prints their sum by category: that is the per-ray listing kept in profiles/.
Issue slots: v_fma_f32-class instructions 1, v_mul_lo_u32 / v_mul_hi_u32 1.7, v_rcp / v_rsq / v_sqrt / v_sin / v_cos 3.5,
v_pk_* 1.75 (tools/microbench/valu_rate.hip)."""
import argparse
import collections
import re

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")
CATS = ["fma/mul/add", "min/max", "cmp", "select", "trans", "int-mul", "int", "cvt", "pk", "lane", "valu-other", "salu", "branch",
        "lds", "vmem", "smem", "wait/nop"]


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_pk_"):
            return "pk"
        if op.startswith(TRANS):
            return "trans"
        if op.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_lo_i32", "v_mul_hi_i32", "v_mad_u64", "v_mad_i64")):
            return "int-mul"
        if op.startswith(("v_cmp", "v_cmpx")):
            return "cmp"
        if op.startswith("v_cndmask"):
            return "select"
        if op.startswith(("v_min", "v_max", "v_med3")):
            return "min/max"
        if op.startswith("v_cvt"):
            return "cvt"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_permlane", "v_bpermute")):
            return "lane"
        if re.match(r"v_(fma|fmac|fmamk|fmaak|mul|add|sub|subrev|mac|mad)_f(32|16|64)", op):
            return "fma/mul/add"
        if re.match(r"v_(add|sub|subrev|and|or|xor|lshl|lshr|ashr|bfe|bfi|mad_u32|mul_u32_u24|mul_i32_i24|add3|or3|xad|and_or|lshl_or|"
                    r"lshl_add|add_lshl|xor3|not|mov|alignbit|perm|addc|subb|sad|ffb|bcnt)", op):
            return "int"
        return "valu-other"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait/nop"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def slots(cat):
    return {"trans": 3.5, "int-mul": 1.7, "pk": 1.75}.get(cat, 1.0)


VALU_CATS = CATS[:11]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--blocks", default="")
    ap.add_argument("--list", action="store_true", help="print the instructions of the chosen blocks")
    a = ap.parse_args()
    lines = open(a.asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and a.kernel in l)
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        m = re.match(r"^\.LBB\d+_(\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        m = re.match(r"^; %bb\.(\d+):", l)
        if m:
            cur = "b" + m.group(1)
            blocks[cur] = []
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        if c:
            blocks[cur].append((op, c, s))
    print(f"kernel {name}: {len(blocks)} blocks")
    hdr = "block".ljust(8) + "".join(c[:9].rjust(10) for c in ["VALU", "slots", "SALU", "LDS", "VMEM", "branch"])
    print(hdr)
    stats = {}
    for b, ins in blocks.items():
        cnt = collections.Counter(c for _, c, _ in ins)
        v = sum(cnt[c] for c in VALU_CATS)
        sl = sum(cnt[c] * slots(c) for c in VALU_CATS)
        stats[b] = cnt
        if ins:
            print(b.ljust(8) + f"{v:10d}{sl:10.1f}{cnt['salu']:10d}{cnt['lds']:10d}{cnt['vmem']:10d}{cnt['branch']:10d}")
    if a.blocks:
        tot = collections.Counter()
        for item in a.blocks.split(","):
            b, _, times = item.partition(":")
            times = int(times) if times else 1
            if b not in stats:
                raise SystemExit(f"no block {b}")
            for c, n in stats[b].items():
                tot[c] += n * times
            if a.list:
                print(f"--- block {b} x{times}")
                for _, c, s in blocks[b]:
                    print(f"    [{c:11s}] {s}")
        print("\nper trip of the chosen blocks:")
        for c in CATS:
            if tot[c]:
                print(f"  {c:12s} {tot[c]:5d}")
        v = sum(tot[c] for c in VALU_CATS)
        sl = sum(tot[c] * slots(c) for c in VALU_CATS)
        print(f"  VALU total   {v:5d}   issue slots {sl:.1f}   SALU {tot['salu']}  LDS {tot['lds']}  branches {tot['branch']}")


if __name__ == "__main__":
    main()
