#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from the -save-temps .s file: tools/isa_loop_stats.py file.s name-substring ..."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
for nm in sys.argv[2:]:
    for m in re.finditer(r'\n(_Z\w*' + re.escape(nm) + r'\w*):', s):
        i = m.start()
        lines = s[i:s.find('s_endpgm', i)].split('\n')
        # blocks of the (largest) innermost loop carry "Loop Header" / "in Loop: Header=BBx_y" comments
        hdrs = Counter()
        for l in lines:
            mm = re.search(r'in Loop: Header=(BB\d+_\d+)', l)
            if mm:
                hdrs[mm.group(1)] += 1
        if not hdrs:
            continue
        hdr = hdrs.most_common(1)[0][0]
        idx = [k for k, l in enumerate(lines) if ('Header=' + hdr + ' ') in l or l.startswith('.L' + hdr + ':')]
        first, last = min(idx), max(idx)
        end = last + 1
        while end < len(lines) and not re.match(r'\.LBB\d+_\d+:', lines[end]):
            end += 1
        loop = lines[first:end]
        ins = [l.strip().split()[0] for l in loop if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
        c = Counter(ins)
        trans = sum(v for k, v in c.items() if k.startswith(('v_exp_f32', 'v_log_f32', 'v_rcp_f32', 'v_rsq', 'v_sqrt')))
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        mfma = sum(v for k, v in c.items() if 'mfma' in k)
        print(m.group(1)[:90])
        print('  longest loop: %d instr, valu %d (trans %d, mfma %d), salu %d, vmem %d, lds %d, branches %d, waitcnt %d' % (
            len(ins), valu, trans, mfma, sum(v for k, v in c.items() if k.startswith('s_')),
            sum(v for k, v in c.items() if k.startswith(('global', 'buffer', 'flat', 'scratch'))),
            sum(v for k, v in c.items() if k.startswith('ds_')), sum(v for k, v in c.items() if k.startswith('s_cbranch')),
            c.get('s_waitcnt', 0)))
        print('  issue cycles (4/valu, 16/trans): %d' % ((valu - trans - mfma) * 4 + trans * 16))
        print('  top:', c.most_common(12))
