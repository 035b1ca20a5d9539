"""Per-stream timeline of the last optimiser step of a rocprofv3 kernel trace: busy time per stream and the
main stream's idle gaps (where it waits for a side stream or for the host)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bertadam' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
per = collections.defaultdict(list)
for r in step:
    per[r['Stream_Id']].append(((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['Kernel_Name']))
for s, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - b for b, e, _ in ks)
    print('stream %s: %4d kernels, busy %.2f ms, first %.2f last %.2f ms' % (s, len(ks), busy / 1e3, ks[0][0] / 1e3, ks[-1][1] / 1e3))
# union coverage: time when at least one kernel runs / when >= 2 run
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']) - t0, 1)); ev.append((int(r['End_Timestamp']) - t0, -1))
ev.sort()
cur, last, cov = 0, 0, collections.defaultdict(float)
for t, d in ev:
    cov[min(cur, 4)] += t - last
    last = t; cur += d
print('concurrency histogram (ms):', {k: round(v / 1e6, 2) for k, v in sorted(cov.items())})
# windows of 1 ms: which concurrency
nb = int(last / 1e6) + 1
sys.path.insert(0, __file__.rsplit('/', 1)[0])
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
    return n[:40]
for s, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
    c = collections.defaultdict(lambda: [0, 0.0])
    for b, e, n in ks:
        c[fam(n)][0] += 1; c[fam(n)][1] += e - b
    print('--- stream', s)
    for k, (n, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:14]:
        print('   %-24s n=%4d %7.3f ms' % (k, n, t / 1e3))
# main-stream gaps
ks = per[max(per, key=lambda k: len(per[k]))]
gaps = sorted(((ks[i + 1][0] - ks[i][1], ks[i][1], ks[i][2][:40], ks[i + 1][2][:40]) for i in range(len(ks) - 1)), reverse=True)
print('main-stream idle total %.2f ms; gaps > 20us: %d' % (sum(g[0] for g in gaps) / 1e3, sum(1 for g in gaps if g[0] > 20)))
for g in gaps[:12]:
    print('   gap %.0f us at %.2f ms after %s before %s' % (g[0], g[1] / 1e3, g[2], g[3]))
