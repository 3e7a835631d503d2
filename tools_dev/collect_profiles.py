"""gpurun_out/<tag> (tools_dev/final_box.sh) + gpurun_out/pmc_<tag> (tools_dev/pmc_bench.sh) -> profiles/round6_*, every file
stamped with the commit it measured:   python tools_dev/collect_profiles.py <tag> <git hash>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, head = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, 'gpurun_out', tag)
prof = os.path.join(ROOT, 'profiles')
box_head = open(os.path.join(src, 'head')).read().strip() if os.path.exists(os.path.join(src, 'head')) else None
assert box_head in (None, head), 'the GPU box measured %s, not %s' % (box_head, head)


def stamp(name):
    with open(os.path.join(prof, os.path.splitext(name)[0] + '.head'), 'w') as f:
        f.write(head + '\n')


# 1. kernel trace summary of the default bench command (one depth map at a time)
stats = os.path.join(src, 'prof', 'bench_kernel_stats.csv')
if os.path.exists(stats):
    with open(stats) as f, open(os.path.join(prof, 'round6_bench_kernel_stats.csv'), 'w') as g:
        g.write('# head %s  (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline '
                '--no-fp32-path --no-power; 36 depth maps incl. warm-up and the eager timing passes)\n' % head)
        g.write(f.read())
    stamp('round6_bench_kernel_stats.csv')
# 2. the PMC passes: the x-pair launches and the table of every kernel
pmc = os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag)
if os.path.exists(os.path.join(pmc, 'summary.json')):
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools_dev', 'make_pmc_profile.py'), tag, 'profiles/round6_pmc_xpair.json'])
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools_dev', 'make_pmc_table.py'), pmc, 'profiles/round6_pmc_kernels.json'])
    for name in ('round6_pmc_xpair.json', 'round6_pmc_kernels.json'):
        p = os.path.join(prof, name)
        d = json.load(open(p))
        d = dict([('head', head)] + list(d.items()))
        json.dump(d, open(p, 'w'), indent=1)
        stamp(name)
# 3. the bench lines and the test verdicts of the same box run
lines = {}
for w in ('default', 'cfg2', 'cfg4', 'cfg5', 'prof'):
    p = os.path.join(src, 'bench_%s.log' % w)
    if os.path.exists(p):
        js = [ln for ln in open(p).read().splitlines() if ln.startswith('{')]
        if js:
            lines[w] = json.loads(js[-1])
res = {'head': head, 'gputests': open(os.path.join(src, 'gputests.log')).read().strip().splitlines()[-1:] if os.path.exists(os.path.join(src, 'gputests.log')) else None,
       'smoke': open(os.path.join(src, 'smoke.log')).read().strip() if os.path.exists(os.path.join(src, 'smoke.log')) else None,
       'bench': lines}
json.dump(res, open(os.path.join(prof, 'round6_final_run.json'), 'w'), indent=1)
stamp('round6_final_run.json')
print('profiles/round6_* written for', head)
for w, ln in lines.items():
    print(w, ln.get('ms_per_step'), ln.get('value'), (ln.get('parity') or {}).get('rel_l1'))
