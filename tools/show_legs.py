import json, sys
d = json.loads(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/legs.json').read().strip().splitlines()[-1])
for k, v in d['legs'].items():
    print(k, v['value'], v['ms_per_step'])
    for kk, vv in v['kernel_ms'].items():
        print('      %-40s %.3f' % (kk, vv))
