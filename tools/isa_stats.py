#!/usr/bin/env python3
"""Instruction counts of chosen kernels in a hipcc -S listing.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/w3.s weath3rb0i_amd/csrc/w3hip.hip
       tools/isa_stats.py /tmp/w3.s k_predict_smallILi8ELb0E k_rank_sortedILi1E"""
import re, sys
s = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r'^(_ZN2w3\w*' + pat + r'\w*): ', s, re.M):
        i = m.end(); j = s.index('.end_amdhsa_kernel', i)
        body = s[i:j]; code = body.split('.section')[0]
        ins = [l.split()[0] for l in code.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        cnt = lambda p: sum(1 for x in ins if x.startswith(p))
        print("%-60s total %5d  valu %5d  salu %5d  ds %4d (add_rtn %d)  global %3d  scratch %d  waitcnt %3d  vgpr %s sgpr %s" % (
            m.group(1)[6:66], len(ins), cnt('v_'), cnt('s_') - cnt('s_waitcnt') - cnt('s_nop'), cnt('ds_'), cnt('ds_add_rtn'), cnt('global_'), cnt('scratch_'),
            cnt('s_waitcnt'), re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1), re.search(r'\.amdhsa_next_free_sgpr (\d+)', body).group(1)))
