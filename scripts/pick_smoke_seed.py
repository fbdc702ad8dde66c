"""Which data seed __graft_entry__.smoke() should use: the fp32 HIP gradients of ResNet18Cifar against the float64 oracle for several
seeds, several runs each (the BatchNorm statistics are summed with atomics, so the low bits differ from run to run; a seed is good when
every run sits at the fp32 noise level, i.e. no ReLU gate is within rounding distance of zero).
    gpurun -- 'python scripts/pick_smoke_seed.py > gpurun_out/smoke_seed.txt'   ->  profiles/r05_smoke_seed.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import __graft_entry__ as G  # noqa: E402

seeds = [int(v) for v in sys.argv[1:]] or list(range(1, 9))
print('data_seed run  logits     loss       worst_grad_elem  grads_vs_f64(HIP)  grads_vs_f64(oracle fp32)  same_gates(HIP)  flipped  margin')
for seed in seeds:
    for run in range(3):
        *_, e = G.smoke_fp32_case(seed)
        print(f"{seed:9d} {run:3d}  {e['logits']:.2e}   {e['loss']:.2e}   {e['worst_grad_element']:.2e}         "
              f"{e['grads_vs_f64_hip']:.2e}           {e['grads_vs_f64_oracle_fp32']:.2e}                   "
              f"{e['grads_vs_f64_same_gates']:.2e}         {e['gates_flipped']:3d}      {e['flipped_margin']:.1e}", flush=True)
