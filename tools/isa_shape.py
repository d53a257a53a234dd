"""Compressed instruction-class sequence of one kernel from a hipcc -S listing (M = MFMA, r = ds_read, G = LDS-DMA, W[..] = s_waitcnt,
|B| = s_barrier, P = s_setprio, v/s = other vector / scalar): python tools/isa_shape.py file.s <mangled-name-substring>"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
i = m.start(); j = s.index('.Lfunc_end', i)
out = []
for l in s[i:j].split('\n'):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        if l.startswith('.LBB'): out.append('\n' + l)
        continue
    op = l.split()[0]
    if op.startswith('v_mfma'): op = 'M'
    elif op.startswith('ds_read'): op = 'r'
    elif op.startswith('ds_write'): op = 'w'
    elif 'load_lds' in op or (op.startswith('buffer_load') and ' lds' in l): op = 'G'
    elif op.startswith('global_load') or op.startswith('buffer_load'): op = 'L'
    elif op.startswith('global_store') or op.startswith('buffer_store'): op = 'S'
    elif op.startswith('s_waitcnt'): op = 'W[' + l.split(None, 1)[1] + ']'
    elif op.startswith('s_barrier'): op = '|B|'
    elif op.startswith('s_setprio'): op = 'P' + l.split()[1]
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): op = '<' + l + '>'
    elif op.startswith('v_'): op = 'v'
    elif op.startswith('s_'): op = 's'
    out.append(op)
txt = ' '.join(out)
for c in 'vsMrwGLS':
    txt = re.sub(r'(?:(?<![\w\[])' + c + r'(?![\w\]]) ?){2,}', lambda m, c=c: f"{c}*{len(m.group(0).split())} ", txt)
print(txt)
