"""Per-step kernel-family breakdown from a rocprofv3 kernel trace (last optimiser step)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bertadam' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6
print('kernels', len(step), 'span ms %.2f' % span, 'busy ms %.2f' % busy)


def fam(n):
    if 'gemm8_kernel' in n:
        cfg = '8'
        if 'ConvColSrc' in n: return 'conv_wgrad_8'
        if 'ConvWeightColSrc' in n: return 'conv_dgrad_8'
        if 'ConvRowSrc' in n: return 'conv_fwd_8'
        if 'PlainSrcIDF16bLb1EEES2_' in n: return 'gemm_NT_fwd_8'
        return 'gemm_NN_dgrad_8'
    if 'gemm_kernel' in n:
        cfg = 'L' if 'Li4ELi4E' in n else 'S'
        if 'ConvColSrc' in n: return 'conv_wgrad_' + cfg
        if 'ConvWeightColSrc' in n: return 'conv_dgrad_' + cfg
        if 'ConvRowSrc' in n: return 'conv_fwd_' + cfg
        if 'PlainSrcIDF16bLb1EEES2_' in n: return 'gemm_NT_fwd_' + cfg
        if 'PlainSrcIDF16bLb0EEES2_' in n: return 'gemm_TN_wgrad_' + cfg
        if 'PlainSrcIDF16bLb1EEENS1_IDF16bLb0' in n: return 'gemm_NN_dgrad_' + cfg
        return 'gemm_other_' + cfg
    for k in ['attn_fwd', 'attn_bwd_dq', 'attn_bwd_dkv', 'ln_fwd', 'ln_bwd', 'bias_act_fwd', 'bias_act_bwd', 'colsum_finish',
              'colsum_partial', 'bertadam', 'hungarian', 'wce', 'bce', 'sumsq', 'ncdhw', 'cast_kernel', 'add_i64', 'copyBuffer',
              'direct_copy', 'CUDAFunctor_add', 'FillFunctor', 'index', 'Cat']:
        if k in n: return k
    return n[:50]


c = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    k = fam(r['Kernel_Name'])
    c[k][0] += 1
    c[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, (n, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:30]:
    print('%-28s n=%4d  %8.3f ms  avg %7.1f us' % (k, n, t / 1e3, t / n))
