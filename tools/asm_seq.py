#!/usr/bin/env python3
"""Instruction-order digest of a kernel's loops from the device assembly: one letter per MFMA (M), LDS read (r) / write (w), global load
(G) / store (S), barrier (|B|), branch (?) and every s_waitcnt as [L<lgkmcnt>V<vmcnt>].  Two round-3 findings came from reading these
strings: conv3p's compiler-scheduled MFMA loop was `r[L0]M` 36 times per tile (every MFMA behind a full LDS round trip), and the weight
gradient's MFMAs each sat in their own exec-masked basic block behind a divergent `if (tap0 + j < TAPS)`.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/k.s csrc/kernels_conv.hip
  python tools/asm_seq.py /tmp/k.s 'conv3p_kernelIDF16_Li3ELi8ELi32ELi4ELi1ELi1ELb0'
"""
import re, sys, collections

def digest(lines):
    seq = []
    for l in lines:
        l = l.strip()
        if l.startswith('v_mfma'): seq.append('M')
        elif l.startswith('ds_read') or l.startswith('ds_load'): seq.append('r')
        elif l.startswith('ds_write') or l.startswith('ds_store'): seq.append('w')
        elif l.startswith('ds_bpermute') or l.startswith('ds_swizzle'): seq.append('x')
        elif l.startswith('s_waitcnt'):
            m = re.search(r'lgkmcnt\((\d+)\)', l); v = re.search(r'vmcnt\((\d+)\)', l)
            seq.append('[' + ('L%s' % m.group(1) if m else '') + ('V%s' % v.group(1) if v else '') + ']')
        elif l.startswith('global_load') or l.startswith('buffer_load'): seq.append('G')
        elif l.startswith('global_store') or l.startswith('buffer_store'): seq.append('S')
        elif l.startswith('global_atomic'): seq.append('A')
        elif l.startswith('s_barrier'): seq.append('|B|')
        elif l.startswith('s_cbranch'): seq.append('?')
    return ''.join(seq)

def main():
    path, pat = sys.argv[1], sys.argv[2]
    s = open(path).read()
    names = [n for n in re.findall(r'^([A-Za-z_][A-Za-z0-9_]*):', s, flags=re.M) if pat in n and not n.startswith('.')]
    for name in names:
        i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
        lines = s[i:j].split('\n')
        ks = [k for k, l in enumerate(lines) if 'Loop Header' in l]
        start = min(ks) if ks else 0
        ops = collections.Counter(l.split()[0] for l in (x.strip() for x in lines[start:]) if l and not l.startswith(('.', ';')))
        print(name); print('  from the first loop header:', digest(lines[start:])[:int(sys.argv[3]) if len(sys.argv) > 3 else 2000])
        print('  static instructions there: %d (mfma %d, valu %d, salu %d, lds %d, vmem %d)' % (
            sum(ops.values()), sum(c for o, c in ops.items() if o.startswith('v_mfma')),
            sum(c for o, c in ops.items() if o.startswith('v_') and not o.startswith('v_mfma')), sum(c for o, c in ops.items() if o.startswith('s_')),
            sum(c for o, c in ops.items() if o.startswith('ds_')), sum(c for o, c in ops.items() if o.startswith(('global_', 'buffer_')))))

if __name__ == '__main__':
    main()
