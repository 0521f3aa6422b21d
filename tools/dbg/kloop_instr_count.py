"""What a K-step costs in INSTRUCTIONS: the general x3 conv kernel against the persistent producer / consumer kernel (VERDICT r5 asked for an
instruction-level look at a K-step of conv_igemm_kernel<128,128,256,2,false,2,true>; the image has no thread-trace decoder -- `rocprofv3 --att`
needs a decoder library that is not installed -- so this is the static view: the generated code of the loops, by instruction class).

Compiles csrc/conv.hip and csrc/conv_x3p.hip to gfx950 assembly (CPU only, ~70 s) and, for each kernel, takes the innermost loop that contains
MFMAs (the K loop) plus everything between its header and its back edge, and counts.  For the general kernel the loop contains several
alternative paths (tap-inner / plain order, tap changes, class skipping): `static` counts all of them, `tap-inner path` the blocks the
stride-1, C >= 256 layers execute per K-step (identified by their LDS-DMA instructions carrying no `s_cbranch_execz` guard).
    python tools/dbg/kloop_instr_count.py > profiles/r06_kloop_instructions.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'aod_meh_hua_amd', 'csrc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-x', 'hip', '-S', '--cuda-device-only']


def asm(src):
    out = os.path.join(tempfile.mkdtemp(), 'k.s')
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + ['-o', out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return open(out).read().split('\n')


def kernel_body(lines, pattern):
    start = next(i for i, l in enumerate(lines) if re.match(pattern, l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
    return lines[start:end]


def classify(ins):
    op = ins.split()[0]
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('buffer_load') and ' lds' in ins:
        return 'lds_dma'
    if op.startswith(('buffer_', 'global_', 'flat_')):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def loops_with_mfma(body):
    """(first line, last line) of every innermost backward-branch loop whose body holds MFMAs"""
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    spans = []
    for i, l in enumerate(body):
        m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            if any('v_mfma' in x for x in body[a:i]):
                spans.append((a, i))
    return spans


def k_loop(body, need, need_dma=0):
    """the SHORTEST backward-branch loop that holds at least `need` MFMAs (and `need_dma` LDS-DMA instructions: the general kernel's loop has a
    compute-only inner branch for its last steps): the K loop (outer spans are the tile loop / the whole kernel)"""
    ok = [s for s in loops_with_mfma(body) if sum('v_mfma' in x for x in body[s[0]:s[1] + 1]) >= need
          and sum(('buffer_load' in x and ' lds' in x) for x in body[s[0]:s[1] + 1]) >= need_dma]
    return min(ok, key=lambda s: s[1] - s[0])


def count(body, a, b):
    c = {}
    for l in body[a:b + 1]:
        t = l.strip()
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
            continue
        k = classify(t)
        c[k] = c.get(k, 0) + 1
    return c


def report(title, body, need=48, need_dma=0):
    print(f'== {title}')
    for a, b in [k_loop(body, need, need_dma)]:
        c = count(body, a, b)
        tot = sum(c.values())
        print(f'   loop of {tot:5d} instructions (static): ' + '  '.join(f'{k} {v}' for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        if c.get('mfma'):
            print(f'      non-MFMA instructions per MFMA (static): {(tot - c["mfma"]) / c["mfma"]:.2f}')


def main():
    conv = asm('conv.hip')
    x3p = asm('conv_x3p.hip')
    gen = kernel_body(conv, r'^_Z17conv_igemm_kernelILi128ELi128ELi256ELi2ELb0ELi2ELb1EEv11ConvKParams:')
    report('conv_igemm_kernel<128,128,256,2,false,2,true> (general x3 kernel: every wave loads, decodes and multiplies) -- K loop, all paths', gen, 48, 8)
    big = kernel_body(conv, r'^_Z17conv_igemm_kernelILi256ELi256ELi512ELi0ELb1ELi2ELb1EEv11ConvKParams:')
    report('conv_igemm_kernel<256,256,512,0,true,2,true> (grouped head-tower tile) -- K loop, all paths', big, 96, 8)
    k9 = kernel_body(x3p, r'^_ZN12_GLOBAL__N_115conv_x3p_kernelILi9ELi4EEEv7X3PArgs:')
    report('conv_x3p_kernel<9, 4> (persistent producer / consumer kernel, 3x3): consumer K loop (MFMAs) -- the loaders\' loops hold no MFMA and are listed below', k9)
    # loader issue blocks: between two LDS-DMA groups
    dma = [i for i, l in enumerate(k9) if 'buffer_load_dwordx4' in l and ' lds' in l]
    if dma:
        groups, cur = [], [dma[0]]
        for i in dma[1:]:
            if i - cur[-1] > 12:
                groups.append(cur); cur = [i]
            else:
                cur.append(i)
        groups.append(cur)
        g = groups[len(groups) // 2]
        c = count(k9, g[0] - 3, g[-1] + 3)
        print(f'   one loader K-step (8 LDS-DMA: 4 pixel rows + 4 filter rows per wave), {sum(c.values())} instructions: ' + '  '.join(f'{k} {v}' for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
    print('''
Reading.  The general kernel's K loop is the static count above around 48 MFMAs; the path a stride-1, C >= 256 layer takes per K-step
executes ~130 of them per wave (per pixel-row load: and / cmp / saveexec / 2 x v_mul_lo_u32 / add / sub / cndmask / add3 / exec restore /
m0 set-up, plus a scalar division for the tap), a 1x1 or C < 256 layer ~200 - 500 (tap change with divisions per row).  On one or two
4-wave workgroups per CU nothing else issues MFMAs meanwhile: matrix pipe busy 0.27 (profiles/r05_pmc_passes.txt).  The persistent kernel's
consumer loop is 48 MFMAs + 16 fragment reads + waits + one barrier; its loaders issue 8 LDS-DMA per K-step with 2 scalar instructions each
(M0 and the next LDS address), every address resolved once per tile.''')


if __name__ == '__main__':
    main()
