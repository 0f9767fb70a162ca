#!/usr/bin/env python3
"""Instruction-class / opcode counts of line ranges of an ISA listing, with multiplicities:
tools/isa_range.py file.s a-b[xN] [c-d[xN] ...] [--top K]"""
import collections
import sys
from isa_hist import classify


def main():
    path = sys.argv[1]
    top = 30
    specs = []
    args = sys.argv[2:]
    while args:
        a = args.pop(0)
        if a == '--top':
            top = int(args.pop(0))
            continue
        mult = 1
        if 'x' in a:
            a, m = a.split('x')
            mult = int(m)
        lo, hi = map(int, a.split('-'))
        specs.append((lo, hi, mult))
    lines = open(path).read().split('\n')
    cls, ops = collections.Counter(), collections.Counter()
    for lo, hi, mult in specs:
        for ln in lines[lo - 1:hi]:
            s = ln.strip()
            if not s or s.startswith((';', '.', '//')) or s.endswith(':') or s.split(';')[0].strip().endswith(':'):
                continue
            op = s.split()[0]
            c = classify(op)
            cls[c] += mult
            ops[(c, op)] += mult
    print(dict(cls), 'total', sum(cls.values()))
    for (c, op), n in ops.most_common(top):
        print(f'{n:6d} {c:5s} {op}')


if __name__ == '__main__':
    main()
