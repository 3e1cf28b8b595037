#!/usr/bin/env python3
"""Instruction mix and register / scratch use of the kernels whose mangled name contains a pattern, from the -save-temps assembly
of a .hip file (cross-compiled: runs without a GPU).   usage: tools/isa_stats.py a3vlm_amd/csrc/a3v_attn.hip w64 [EXTRA flags]"""
import re, subprocess, sys, os, tempfile
from collections import Counter
src, pat = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
tmp = tempfile.mkdtemp()
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-S", "--cuda-device-only", src,
       "-o", os.path.join(tmp, "k.s")] + extra
subprocess.run(cmd, check=True)
s = open(os.path.join(tmp, "k.s")).read()
keys = ['v_mfma_f32_32x32x16_bf16', 'v_mfma_f32_16x16x32_bf16', 'v_exp_f32', 'v_accvgpr_read_b32', 'v_accvgpr_write_b32', 'scratch_load_dword',
        'scratch_store_dword', 'scratch_load_dwordx4', 'scratch_store_dwordx4', 'ds_read_b128', 'ds_read_b64', 'v_permlane32_swap_b32',
        'v_cvt_pk_bf16_f32', 'v_max3_f32', 'v_max_f32', 'v_fma_f32', 'v_add_f32', 'v_pk_add_f32', 'v_pk_mul_f32', 'v_pk_fma_f32', 'v_mul_f32', 's_nop',
        'v_mov_b32', 'v_xor_b32', 'v_add_u32', 'ds_bpermute_b32', 'buffer_load_dwordx4', 's_barrier', 's_waitcnt', 'v_cndmask_b32']
for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':')]
    c = Counter(i.split()[0] for i in ins)
    print(name[:90], 'instructions', len(ins))
    print('   ', {k: c[k] for k in keys if c.get(k)})
    for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_private_segment_fixed_size', '.amdhsa_group_segment_fixed_size'):
        mm = re.search(re.escape(k) + r'\s+(\d+)', body + s[m.end():m.end() + 10])
    tail = s[m.start():]
    for k in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_private_segment_fixed_size'):
        mm = re.search(re.escape(k) + r'\s+(\d+)', tail)
        if mm: print('   ', k, mm.group(1))
if '--keep' in extra or os.environ.get('KEEP_S'):
    print(os.path.join(tmp, 'k.s'))
