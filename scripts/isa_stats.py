#!/usr/bin/env python3
"""Instruction mix / register use of kernels in a hipcc -save-temps .s file:  isa_stats.py file.s pattern [pattern ...]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
for kn in sys.argv[2:]:
    for m in re.finditer(r'^(_Z\w*' + kn + r'\w*):', s, re.M):
        name = m.group(1)
        i = m.start()
        j = s.index('s_endpgm', i)
        c = collections.Counter()
        for line in s[i:j].split('\n'):
            line = line.strip()
            if not line or line.startswith(('.', ';')) or line.endswith(':'):
                continue
            c[line.split()[0]] += 1
        k = s.find('.name:           ' + name + '\n')
        meta = s[s.rindex('- .agpr_count', 0, k):k + 600] if k > 0 else ''
        def g(key):
            r = re.search(key + r':\s+(\d+)', meta)
            return r.group(1) if r else '?'
        print(name[:70], 'VALU', sum(v for k2, v in c.items() if k2.startswith('v_')), 'vgpr', g(r'\.vgpr_count'), 'spill', g(r'\.vgpr_spill_count'),
              'scratch', g(r'\.private_segment_fixed_size'), 'lds', g(r'\.group_segment_fixed_size'))
        print('    ', c.most_common(10))
