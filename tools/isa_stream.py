"""One character per instruction of a kernel's main loop (hipcc -S output): M mfma, d ds_read, D ds_write, w s_waitcnt lgkmcnt,
W s_waitcnt vmcnt, n s_nop, G global/buffer memory, B barrier, s other scalar, . vector ALU.
usage: python tools/isa_stream.py file.s <kernel name substring>"""
import re
import sys
import textwrap

lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
body = lines[start:end]
labels = {l.split(':')[0].strip(): i for i, l in enumerate(body) if re.match(r'^\.?[A-Za-z_0-9$.]+:', l.strip())}
best = (0, 0, len(body))
for i, l in enumerate(body):
    m = re.match(r'\s*s_cbranch_\w+\s+(\S+)', l) or re.match(r'\s*s_branch\s+(\S+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
out = []
for l in body[best[1]:best[2] + 1]:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): out.append('M')
    elif op.startswith('ds_read'): out.append('d')
    elif op.startswith('ds_'): out.append('D')
    elif op.startswith('s_waitcnt'): out.append('w' if 'lgkm' in l else 'W')
    elif op.startswith('s_nop'): out.append('n')
    elif op.startswith('s_barrier'): out.append('B')
    elif op.startswith('v_'): out.append('.')
    elif op.startswith('global_') or op.startswith('buffer_'): out.append('G')
    else: out.append('s')
print('\n'.join(textwrap.wrap(''.join(out), 150)))
