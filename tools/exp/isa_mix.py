"""Instruction mix of one kernel in a hipcc -S listing: python tools/exp/isa_mix.py file.s <mangled-name-substring>"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
cnt = collections.Counter()
detail = collections.Counter()
for l in lines[start:end]:
    t = l.strip().split()
    if not t or t[0].endswith(':') or t[0].startswith(('.', ';')):
        continue
    op = t[0]
    detail[op] += 1
    if op.startswith('v_mfma'): cnt['mfma'] += 1
    elif op.startswith('v_'): cnt['valu'] += 1
    elif op.startswith('ds_'): cnt['lds'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cnt['vmem'] += 1
    elif op.startswith('s_waitcnt'): cnt['waitcnt'] += 1
    elif op.startswith('s_nop'): cnt['nop'] += 1
    elif op.startswith('s_'): cnt['salu'] += 1
    else: cnt['other'] += 1
print(dict(cnt), 'total', sum(cnt.values()))
if len(sys.argv) > 3:
    for op, c in detail.most_common(int(sys.argv[3])): print(f'{c:6d} {op}')
