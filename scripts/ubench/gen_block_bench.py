#!/usr/bin/env python3
"""Generate a micro-benchmark from a basic block of compiled ISA: the block (a text file of instructions, one per line) is replayed
N_ITER times as ONE inline-asm statement with its original registers, whole and with classes of instructions filtered out, at a given
number of waves per CU.  Timing only: registers hold whatever they hold.  Usage: gen_block_bench.py block.txt out.hip"""
import re, sys
ins = [l.strip() for l in open(sys.argv[1]) if l.strip()]
ins0 = ins
ins = [l for l in ins if not l.startswith('s_cbranch') and not l.startswith('s_branch') and not l.startswith('s_cmp') and not l.startswith('s_and_b64') and not l.startswith('s_or_b32') and not l.startswith('s_mov') and not l.startswith('s_add') and not l.startswith('ds_') and not l.startswith('s_waitcnt') and not l.startswith('s_load')]
def variant(pred): return [l for l in ins if pred(l)]
tiny = lambda l: l.startswith(('v_lshl_add_u32', 'v_min3_u32', 'v_min_u32', 'v_cmp_'))
pk = lambda l: l.startswith('v_pk_')
variants = {
    'whole block': variant(lambda l: True),
    'without the tiny test': variant(lambda l: not tiny(l)),
    'tiny test only': variant(tiny),
    'packed only': variant(pk),
    'without med3': variant(lambda l: not l.startswith('v_med3')),
    'without dpp': variant(lambda l: 'dpp' not in l),
    'without s_nop': variant(lambda l: not l.startswith('s_nop')),
    'packed + tiny': variant(lambda l: pk(l) or tiny(l)),
    'packed + med3': variant(lambda l: pk(l) or l.startswith('v_med3')),
    'dpp -> plain v_mov': [re.sub(r'_dpp (v\d+), (v\d+).*', r'_e32 \1, \2', l) if 'dpp' in l else l for l in ins],
    'wave_sh -> row_sh': [l.replace('wave_shl', 'row_shl').replace('wave_shr', 'row_shr') for l in ins],
    'wave_sh -> quad_perm': [re.sub(r'wave_sh[lr]:1', 'quad_perm:[1,2,3,0]', l) for l in ins],
    'dpp first': [l for l in ins if 'dpp' in l] + [l for l in ins if 'dpp' not in l],
    'dpp last': [l for l in ins if 'dpp' not in l] + [l for l in ins if 'dpp' in l],
}
used = set()
for l in ins:
    for m in re.finditer(r'v\[?(\d+)(?::(\d+))?\]?', l):
        used.update(range(int(m.group(1)), int(m.group(2) or m.group(1)) + 1))
free = [r for r in range(max(used)) if r not in used] + [max(used) + 1, max(used) + 2, max(used) + 3]
dpps = [l for l in ins if 'dpp' in l]
nodpp = [l for l in ins if 'dpp' not in l]
# the lane shifts as LDS crossbar permutes (ds_bpermute_b32: no LDS memory, no VALU): issued at the top, waited for at the bottom --
# in the kernel they would be issued a sweep ahead and their latency covered by the wait for the neighbours' rows
bperm = ['ds_bpermute_b32 v%d, v%d, %s' % (free[1 + k % 2], free[0], re.search(r'_dpp v\d+, (v\d+)', l).group(1)) for k, l in enumerate(dpps)]
variants['dpp -> ds_bpermute (top), wait (bottom)'] = bperm + nodpp + ['s_waitcnt lgkmcnt(0)']
variants['dpp -> ds_bpermute + wait (top)'] = bperm + ['s_waitcnt lgkmcnt(0)'] + nodpp
maxv = 0
for l in [x for b in variants.values() for x in b]:
    for m in re.finditer(r'v\[?(\d+)(?::(\d+))?\]?', l):
        maxv = max(maxv, int(m.group(2) or m.group(1)))
clob = ', '.join('"v%d"' % i for i in range(maxv + 1)) + ', "vcc"'
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#define N_ITER 2000', '#define MAXW %d' % (16 if maxv < 128 else 12 if maxv < 168 else 8)]
for k, (name, body) in enumerate(variants.items()):
    s = '\\n\\t'.join(body)
    out.append('__global__ void k%d(float *o) { for (int it = 0; it < N_ITER; it++) asm volatile("%s" ::: %s); o[0] = 0; }' % (k, s, clob))
out.append('''template <typename K> void run(K kern, const char *name, int n, float *o) {
    printf("%-26s %3d instr:", name, n);
    for (int cus : {1, 256}) for (int wpc : {4, 8, 12, 16}) { if (wpc > MAXW) continue;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<cus, 64 * wpc>>>(o); (void)hipEventRecord(e0); kern<<<cus, 64 * wpc>>>(o); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; hipError_t er = hipEventElapsedTime(&ms, e0, e1); if (er != hipSuccess || hipGetLastError() != hipSuccess) { printf("  %dcu/%2dw  fail", cus, wpc); continue; }
        printf("  %dcu/%2dw %5.2f", cus, wpc, ms * 1e-3 * 2.4e9 / ((double)N_ITER * n * wpc / 4));
    }
    printf("\\n");
}
int main() { setvbuf(stdout, nullptr, _IONBF, 0); float *o; hipError_t e_ = hipMalloc(&o, 4096); printf("malloc %d\\n", (int)e_); printf("cycles per wave-instruction per SIMD @2.4 GHz nominal; CUs busy / waves per CU\\n");''')
for k, (name, body) in enumerate(variants.items()):
    out.append('    run(k%d, "%s", %d, o);' % (k, name, len(body)))
out.append('    return 0; }')
open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
