#!/usr/bin/env python3
"""Instruction-class counts of one kernel per region between s_barrier instructions (static listing order).
tools/isa_regions.py file.s kernel-substring"""
import collections
import sys
from isa_hist import kernel_lines, classify


def main():
    path, key = sys.argv[1], sys.argv[2]
    regions, cur, movs = [], collections.Counter(), collections.Counter()
    for ln in kernel_lines(path, key):
        s = ln.strip()
        if not s or s.startswith((';', '.', '//')):
            continue
        if s.endswith(':'):
            cur['label'] += 1
            continue
        op = s.split()[0]
        if op == 's_barrier':
            regions.append(cur)
            cur = collections.Counter()
            continue
        c = classify(op)
        cur[c] += 1
        if op in ('v_mov_b32_e32', 'v_cvt_pk_bf16_f32', 'v_pk_add_f32', 'v_lshl_add_u64', 's_nop', 'v_sub_f32_e32', 'v_lshlrev_b32_e32', 'v_and_b32_e32', 'v_max_f32_e32', 'global_store_dwordx4', 'global_load_dwordx4', 'ds_read_b128', 'ds_write_b128'):
            cur[op] += 1
    regions.append(cur)
    for i, r in enumerate(regions):
        print(i, ' '.join(f'{k}={v}' for k, v in sorted(r.items())))


if __name__ == '__main__':
    main()
