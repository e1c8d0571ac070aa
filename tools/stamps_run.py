import sys
sys.path.insert(0, '/root/repo')
import secedo_amd, torch
from secedo_amd.synth import synth_config, CONFIGS
name = sys.argv[1]
n = CONFIGS[name][0]
p = synth_config(name)
with secedo_amd.SimilarityMatrixPlan(0) as plan:
    plan.prepare(p, n, 1000, None, 8)
    acc = plan.new_acc()
    for _ in range(2):
        plan.accumulate(acc, 0.01, 0.5, 0.01)
        torch.cuda.synchronize()
    print(name, "kernel ms", plan.last_accumulate_ms(), plan.last_counts())
