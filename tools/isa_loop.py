"""Print the sync-relevant instruction sequence (waits, barriers, DMA, LDS, MFMA counts) of a kernel's ISA.
usage: isa_loop.py file.s mangled-substring"""
import re, sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r'^(_Z\w+):', s, re.M) if sys.argv[2] in m.group(1)]
for name in names[:1]:
    i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
    body = s[i:j].split('\n')
    print(name)
    run = {}
    def flush():
        if run:
            print('      ', ' '.join(f'{k}x{v}' for k, v in run.items()))
            run.clear()
    for k, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((';', '.')) and not re.match(r'\.LBB', t):
            continue
        op = t.split()[0]
        if re.match(r'\.LBB', t) or op.startswith(('s_waitcnt', 's_barrier', 's_cbranch', 's_branch')):
            flush(); print(k, t[:90])
        else:
            key = ('mfma' if 'mfma' in op else 'dma' if 'load_lds' in op else 'ds' if op.startswith('ds_') else
                   'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'valu' if op.startswith('v_') else 'salu')
            run[key] = run.get(key, 0) + 1
    flush()
