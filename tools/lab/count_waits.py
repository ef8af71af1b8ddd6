"""Lab: per kernel of a device assembly listing, the number of vector-memory loads and of s_waitcnt vmcnt waits (full
waits = vmcnt(0)).  Many full waits for few loads = load -> wait -> use chains the compiler could not batch (conditional
per-element loads): each one is an L2 round trip on the kernel's critical path.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I view_neti_amd/csrc -o /tmp/k.s view_neti_amd/csrc/norms.hip
    python tools/lab/count_waits.py /tmp/k.s"""
import re, sys
txt=open(sys.argv[1]).read()
idx=[(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN12_GLOBAL__N_1\S+?):', txt, re.M)]
for i,(pos,name) in enumerate(idx):
    end = txt.find('s_endpgm', pos)
    body = txt[pos:end]
    lines=[l for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.',';'))]
    loads=sum(1 for l in lines if re.search(r'\b(global_load|buffer_load|scratch_load)', l))
    waits=[l.strip() for l in lines if 's_waitcnt' in l and 'vmcnt' in l]
    print(f"{name[18:88]:70s} instrs {len(lines):5d} vmem loads {loads:3d} vm waits {len(waits):3d} vmcnt(0) {sum('vmcnt(0)' in w for w in waits)}")
