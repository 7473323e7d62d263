#!/usr/bin/env python3
"""Instruction histogram of kernels in a hipcc -S listing: tools/isa_hist.py file.s substr [substr...]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    for m in re.finditer(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % re.escape(name), txt, re.S | re.M):
        c = collections.Counter()
        for line in m.group(2).split('\n'):
            line = line.strip()
            if not line or line[0] in ';.' or line.split()[0].endswith(':'): continue
            c[line.split()[0]] += 1
        print(m.group(1), sum(c.values()))
        print('   ', ' '.join(f'{k}={v}' for k, v in c.most_common(24)))
