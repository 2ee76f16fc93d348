#!/usr/bin/env python3
"""ISA-level variants of ONE kernel of a device assembly (tools/isa_variants/setup.sh): same instructions, same registers, only wait states added — or only
the kernel descriptor changed (profiles/r05_miscompile.md).
usage: mk_variant.py <variant> : writes rb_graph-hip-amdgcn-amd-amdhsa-gfx950.s from dev_orig.s"""
import re, sys
v = sys.argv[1]
L = open('dev_orig.s').read().split('\n')
a = next(i for i, l in enumerate(L) if re.match(r'^_ZN.*k_pairs_insert_runtime_branch.*:', l))
b = next(i for i in range(a, len(L)) if 's_endpgm' in L[i])
out = L[:a]
n = 0
for l in L[a:b + 1]:
    out.append(l)
    t = l.strip()
    add = False
    if v == 'valu_all': add = t.startswith('v_')
    elif v == 'valu_sgpr': add = bool(re.match(r'v_cmp\w*_e64 s\[|v_mad_u64_u32|v_(add|sub|subb|addc)\w*_co\w* |v_readlane|v_readfirstlane', t))
    elif v == 'salu_all': add = t.startswith('s_') and not t.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_waitcnt', 's_nop'))
    elif v == 'vmem_all': add = t.startswith(('global_', 'flat_', 'buffer_'))
    elif v == 'all': add = (t.startswith(('v_', 'global_')) or (t.startswith('s_') and not t.startswith(('s_cbranch', 's_branch', 's_endpgm'))))
    elif v == 'none': add = False
    elif v == 'valu_all2': add = 2 if t.startswith('v_') else False
    elif v == 'mad64': add = t.startswith('v_mad_u64_u32')
    elif v == 'mul32': add = t.startswith(('v_mul_lo_u32', 'v_mul_hi_u32'))
    elif v == 'cmp64': add = bool(re.match(r'v_cmp\w*_e64 s\[', t))
    elif v == 'cmp_all': add = t.startswith('v_cmp')
    elif v == 'sh64': add = t.startswith(('v_lshlrev_b64', 'v_lshrrev_b64', 'v_ashrrev_i64'))
    elif v == 'add64': add = t.startswith('v_lshl_add_u64')
    elif v == 'cnd': add = t.startswith('v_cndmask')
    elif v == 'co': add = bool(re.match(r'v_(add|sub|subb|addc|subrev)\w*_co\w* ', t))
    elif v == 'before_sh64': add = False
    elif v == 'xor_or': add = t.startswith(('v_xor_b32', 'v_or_b32', 'v_and_b32'))
    if add:
        for _ in range(int(add)): out.append('\ts_nop 7')
        n += 1
if v.startswith('swap'):          # swapA_B: rename two single registers throughout the kernel (same instructions, same count: only WHICH value lives where)
    ra, rb = re.match(r'swap(\d+)_(\d+)', v).groups()
    k0 = len(L[:a])
    def sw(line):
        code, sep, cmt = line.partition(';')
        code = re.sub(r'\bv(%s|%s)\b' % (ra, rb), lambda m: 'v' + (rb if m.group(1) == ra else ra), code)
        return code + sep + cmt
    out[k0:] = [sw(l) for l in out[k0:]]
tail = L[b + 1:]
# descriptor-only variants: no instruction changes, the kernel descriptor asks for more registers
for vv in (v.split('+') if v.startswith(('sgpr', 'vgpr', 'accum')) else []):
    kind, val = re.match(r'([a-z]+)(\d+)', vv).groups()
    key = {'sgpr': '.amdhsa_next_free_sgpr', 'vgpr': '.amdhsa_next_free_vgpr', 'accum': '.amdhsa_accum_offset'}[kind]
    hit = 0
    seen_kernel = False
    for i, l in enumerate(tail):
        if '.amdhsa_kernel' in l: seen_kernel = 'k_pairs_insert_runtime_branch' in l
        if seen_kernel and l.strip().startswith(key + ' '):
            tail[i] = '\t\t' + key + ' ' + val; hit += 1
        if '.end_amdhsa_kernel' in l: seen_kernel = False
    print('descriptor lines changed', hit)
out += tail
open('rb_graph-hip-amdgcn-amd-amdhsa-gfx950.s', 'w').write('\n'.join(out))
print(v, 'kernel lines', b - a, 'nops added', n)
