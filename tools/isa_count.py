"""Static instruction counts per kernel from a hipcc -S dump (tools/isa_count.py file.s [kernel substring])."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'^(_Z\w*k_\w+):[^\n]*\n(.*?)\n\s*\.section\s+\.rodata', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    valu = sum(1 for i in ins if i.startswith('v_'))
    ds = sum(1 for i in ins if i.startswith('ds_'))
    vmem = sum(1 for i in ins if i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')))
    sal = sum(1 for i in ins if i.startswith('s_'))
    tail = s[m.end():m.end() + 6000]
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', tail)
    sp = re.search(r'ScratchSize: (\d+)', tail)
    print(f"{name[:70]:70s} total {len(ins):5d} valu {valu:5d} ds {ds:4d} vmem {vmem:4d} salu {sal:5d} vgpr {vg.group(1) if vg else '?'} scratch {sp.group(1) if sp else '?'}")
