#!/usr/bin/env python3
"""tools/isa_stats.py <asm.s> <kernel-substring> [--loops]: register/LDS usage of a kernel and its MFMA loops."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    if pat not in m.group(1):
        continue
    body = m.group(2)
    print(m.group(1))
    for key in ['next_free_vgpr', 'next_free_sgpr', 'accum_offset', 'group_segment_fixed_size', 'private_segment_fixed_size']:
        for l in body.split('\n'):
            if key in l:
                print('  ', l.strip())
    print('   mfma', body.count('v_mfma'), 'accvgpr', body.count('v_accvgpr'), 'scratch', body.count('scratch_'), 'lines', body.count('\n'))
    if '--loops' in sys.argv:
        lines = body.split('\n')
        for i, l in enumerate(lines):
            if 'Inner Loop Header' in l and any('v_mfma' in x for x in lines[i:i + 40]):
                j = i
                while j < len(lines) and 's_cbranch' not in lines[j]:
                    j += 1
                print('\n'.join(lines[i - 1:j + 1]))
                print('-----')
