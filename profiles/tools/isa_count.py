#!/usr/bin/env python3
"""Instruction mix per kernel from a hipcc -save-temps gfx950 .s file (static counts; loops counted once).
usage: isa_count.py file.s [name-substring ...]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split('\n')
    pats = sys.argv[2:]
    starts = [(i, l.split(':')[0]) for i, l in enumerate(txt) if re.match(r'^_Z\S+:', l)]
    starts.append((len(txt), None))
    for (i0, name), (i1, _) in zip(starts, starts[1:]):
        if pats and not any(p in name for p in pats):
            continue
        ins = []
        for l in txt[i0:i1]:
            if not l.startswith('\t'):
                continue
            t = l.strip()
            if not t or t[0] in '.;':
                continue
            ins.append(t.split()[0])
            if t.startswith('s_endpgm'):
                break
        c = collections.Counter(ins)
        g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        f64 = sum(v for k, v in c.items() if 'f64' in k)
        print(f"{name[:70]}\n   total {len(ins)} valu {g('v_')} f64 {f64} rcp {g('v_rcp_f64')} rsq {g('v_rsq_f64')} sqrt {g('v_sqrt_f64')} div_scale {g('v_div_scale')} "
              f"fma {g('v_fma_f64')} mul {g('v_mul_f64')} add {g('v_add_f64')} cmp {g('v_cmp')} cndmask {g('v_cndmask')} mov {g('v_mov')} "
              f"accvgpr {g('v_accvgpr')} gload {g('global_load')} gstore {g('global_store')} scratch {g('scratch_')} ds {g('ds_')} salu {g('s_')} waitcnt {c['s_waitcnt']}")


main()
