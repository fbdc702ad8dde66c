"""Training-throughput bench of the SimpleAICV DDP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model resnet50|vit_base_patch16|sam_b_encoder|resnet50_detr|resnet50_detr_config|resnet50_retinanet|resnet50_fcos|sam_b]
                    [--batch B] [--no-cpu-baseline] [--no-secondary] [--eager]

A step = forward + loss + backward + gradient all-reduce + optimizer step of one per-GPU batch of synthetic data that
is resident in HBM before timing starts.  `--gpus N` with N > 1 starts N ranks itself (one process per GPU, RCCL via
torch.distributed backend "nccl"), as the reference does with torchrun (00.classification_training/imagenet/resnet50/
train.sh, tools/train_classification_model.py:52-66); under torchrun (RANK / WORLD_SIZE already set) it joins instead.

The classification workloads are driven through the product path the reference user takes: the benchmark copy of the
reference train_config.py (00.classification_training/imagenet/<model>/train_config.py) -> tools.utils.build_optimizer
-> build_training_mode -> tools.scripts.train_classification over a loader of K device-resident batches.  With one GPU
the loop runs each iteration as a replayed hipGraph (config.use_step_graph, engine.StepGraph); `--eager` turns that off.

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` / `ms_per_step` are the MEDIAN of several
timed windows of exactly K steps each (each bracketed by barrier + synchronize; max over ranks), enough windows for
>= 5 s of timed GPU work; all windows are listed.  Extra objects:
  roofline     -- the dominant kernel (implicit-GEMM conv, MFMA-bound) priced from HIP events around every one of its
                  launches in an eager pass of the same steps right after the timed windows (events cannot bracket
                  kernels inside a replayed graph);
  secondary    -- ViT-B/16 b256 (the metric names both models) when the default ResNet-50 line is requested;
  cpu_baseline -- the reference's own modules (when /root/reference is importable) or the CPU oracle restatement,
                  timed on this box's host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import importlib.util
import json
import logging
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# r06: ROCm's graph packet capture (hipGraph replays as pre-recorded AQL packets) gave wrong results in two captured steps (the SAM
# step; a deterministic ResNet-50 step -- DESIGN.md section 3k) and buys nothing measurable (ResNet-50 20.42 / 20.41 ms, ViT-B 39.02 /
# 39.02 ms on / off): every captured step takes the conventional replay path.  The package sets the same default when it is imported;
# here too because the runtime reads the switch before its first call and torch is imported below.
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
# SURVEY.md section 8(d); DETR: 189.1 GFLOP fwd at 800x1344, scaled to the 800x1333 content of the padded canvas
TRAIN_GFLOP_PER_IMG = {'resnet50': 24.54, 'vit_base_patch16': 105.38, 'sam_b_encoder': 3 * 972.1,
                       'resnet50_detr': 3 * 189.1 * (800 * 1333) / (800 * 1344)}
IMAGE_SIZE = {'sam_b_encoder': 1024, 'resnet50_detr': 1333}
CONFIG_DIR = {'resnet50': '00.classification_training/imagenet/resnet50',
              'vit_base_patch16': '00.classification_training/imagenet/vit_base_patch16_for_self_train_mae_pretrain',
              # the reference's own DETR / SAM training configurations (1024^2 canvases), through their loops
              'resnet50_detr_config': '03.detection_training/coco/res50_detr_yoloresize1024',
              'resnet50_retinanet': '03.detection_training/coco/res50_retinanet_yoloresize1024',
              'resnet50_fcos': '03.detection_training/coco/res50_fcos_yoloresize1024',
              'sam_b': '13.interactive_segmentation_training/13.1.sam_segmentation_training/sam_b_training'}
LOOP_MODELS = {'resnet50_detr_config': ('tools.scripts.train_detection', 8, 1024, 3 * 189.1 * (768 * 1024) / (800 * 1344)),
               # dense detectors through the same loop (reference per-GPU batch 32 / 8 = 4); no FLOP figure is published for them
               'resnet50_retinanet': ('tools.scripts.train_detection', 4, 1024, 0.0),
               'resnet50_fcos': ('tools.scripts.train_detection', 4, 1024, 0.0),
               # full SAM step: one encoder pass (972.1 GFLOP fwd) + 1 + decoder_iters light decoder passes
               'sam_b': ('tools.interactive_segmentation_scripts.train_sam_segmentation', 8, 1024, 3 * 972.1)}
# HBM bytes per launch per kernel family, from separate rocprofv3 --pmc passes of the same command (scripts/gpu_r05.sh pmc:<model>)
PMC_FILES = {'resnet50': ['profiles/r06_pmc_hbm_traffic.json', 'profiles/r05_pmc_hbm_traffic.json', 'profiles/r04_pmc_hbm_traffic.json', 'profiles/r03_pmc_hbm_traffic.json', 'profiles/r02_pmc_hbm_traffic.json'],
             'vit_base_patch16': ['profiles/r06_pmc_hbm_traffic_vit_base_patch16.json', 'profiles/r05_pmc_hbm_traffic_vit_base_patch16.json', 'profiles/r04_pmc_hbm_traffic_vit_base_patch16.json']}
# what each bracketed family is in the rocprofv3 kernel lists
FAMILY_KERNELS = {'igemm_nt': 'igemm_nt1_kernel (implicit-GEMM conv / linear: forward + data gradient)',
                  'igemm_tn': 'igemm_tn_dma_kernel (weight gradient)',
                  'bn_act_fwd': 'bn_act_fwd_kernel (BatchNorm apply + residual + ReLU)',
                  'bn_act_bwd': 'bn_bwd_apply_kernel (+ bn_bwd_reduce_kernel where the reduction is not fused into the data gradient)',
                  'layernorm_fwd': 'layernorm_fwd_kernel', 'layernorm_bwd': 'layernorm_bwd_kernel (+ colreduce)',
                  'attention_fwd': 'sa_fwd_kernel / attention_fwd_kernel', 'attention_bwd': 'attention_bwd_kernel / sa_bwd_dq + sa_bwd_dkv'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='resnet50')
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default 256; 8 for sam_b_encoder / resnet50_detr)')
    ap.add_argument('--min-gpu-seconds', type=float, default=5.0, help='timed windows are repeated until this much timed work ran')
    ap.add_argument('--max-windows', type=int, default=15)
    ap.add_argument('--eager', action='store_true', help='no step graph (one kernel launch per kernel from Python)')
    ap.add_argument('--graph', action='store_true', help='force the step graph also with several ranks')
    ap.add_argument('--deterministic', action='store_true',
                    help='fixed-order BatchNorm statistics, as the entry scripts run after set_seed() (default: fp32 atomics)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--kernel-breakdown', action='store_true', help='(default since r04; kept for old command lines)')
    ap.add_argument('--dominant-only', action='store_true',
                    help='bracket only the dominant kernel with HIP events in the eager pricing pass (default: every instrumented family)')
    ap.add_argument('--no-sam', action='store_true', help='skip the sam_b_encoder object of the default line')
    ap.add_argument('--no-power', action='store_true', help='skip the rocm-smi power / clock sample after the timed windows')
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ launch
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn(args):
    """One worker process per GPU with the torchrun environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*); rank 0's
    stdout (the JSON line) passes through.  Any rank failing fails the bench."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), SAICV_BENCH_SPAWNED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3600
    while procs and time.time() < deadline:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0:
                rc = rc or code
                for q in procs:         # a dead rank leaves the others waiting in a collective
                    q.terminate()
        time.sleep(0.2)
    for p in procs:
        p.kill()
        rc = rc or 1
    return rc


# ------------------------------------------------------------------------------------------ workloads
def load_config(model_name):
    path = os.path.join(ROOT, CONFIG_DIR[model_name], 'train_config.py')
    spec = importlib.util.spec_from_file_location(f'bench_train_config_{model_name}', path)
    mod = importlib.util.module_from_spec(spec)
    # the config builds its model while it is imported -- BEFORE the entry script's set_seed(), in the reference too (its weights
    # are a fresh draw per process).  Seeded here so that two bench processes start from the same weights and `final_loss` compares.
    import torch
    torch.manual_seed(0)
    spec.loader.exec_module(mod)
    return mod.config, os.path.relpath(path, ROOT)


class DeviceLoader(list):
    """K references to batches already resident in HBM; len(dataset) // batch_size = iterations per epoch
    (tools/scripts.py train_classification reads it like the reference, :137)."""

    def __init__(self, batches, global_batch, iters_per_epoch):
        super().__init__(batches)
        self.dataset = range(global_batch * iters_per_epoch)


def classification_workload(name, args, world, rank, device, use_graph):
    """The product path: train_config.py -> build_optimizer -> build_training_mode -> train_classification."""
    import numpy as np
    import torch
    from simpleaicv_pytorch_training_examples_amd.tools import scripts, utils
    config, cfg_path = load_config(name)
    _CONFIGS[name] = config
    batch = args.batch or 256
    utils.set_seed(config.seed)
    np.random.seed(config.seed + rank)
    config.local_rank, config.gpus_num, config.group = device.index, world, None
    config.batch_size = batch * world                      # weak scaling: the per-GPU batch is fixed
    config.use_step_graph = use_graph
    config.host_sync_lag = 2
    config.print_interval = 10 ** 9
    model = config.model.to(device)
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, config.ema_model, config.scaler = utils.build_training_mode(config, model)
    # one per-GPU batch through the config's own collater (ResNet: hard labels; ViT: Mixup / CutMix soft labels),
    # moved to the device once: inputs are resident in HBM when the timed region starts
    samples = [config.train_dataset[rank * batch + i] for i in range(batch)]
    data = config.train_collater(samples)
    data = {'image': data['image'].to(device), 'label': data['label'].to(device)}
    iters_per_epoch = len(config.train_dataset) // config.batch_size
    logger = logging.getLogger('saicv_bench')
    logger.addHandler(logging.NullHandler())
    logger.propagate = False
    state = {'loss': float('nan')}

    def run(k):
        state['loss'] = scripts.train_classification(DeviceLoader([data] * k, config.batch_size, iters_per_epoch), model,
                                                     config.train_criterion, optimizer, scheduler, 1, logger, config)
        graphs = getattr(config, '_saicv_step_graphs', None)
        state['step_graph'] = max(graphs.values(), key=lambda g: g.replays) if graphs else None
        state['step_graphs'] = {'captured': sum(g.graph is not None for g in graphs.values()), 'replays': sum(g.replays for g in graphs.values())} if graphs else None

    info = {'config_file': cfg_path, 'loop': 'tools.scripts.train_classification', 'optimizer': config.optimizer[0],
            'param_groups': len(optimizer.param_groups)}
    return run, model, config.scaler, state, info, batch, 224


def loop_workload(name, args, world, rank, device, use_graph=False):
    """DETR / full SAM through the reference's own config and loop.  DETR (r05) runs as ONE captured step when `use_graph`: its
    Hungarian assignment is a device kernel (saicv_detr_assign).  The SAM loop (r06) is captured too: one graph per drawn prompt
    combination (the reference config draws point-only or box-only prompts: two graphs, each captured after three eager steps) when
    its config asks for it (SAICV_SAM_GRAPH=1; off by default, see tools/interactive_segmentation_scripts.py)."""
    import numpy as np
    import torch
    from simpleaicv_pytorch_training_examples_amd.tools import interactive_segmentation_scripts, scripts, utils
    loop_name, default_batch, size, _ = LOOP_MODELS[name]
    config, cfg_path = load_config(name)
    _CONFIGS[name] = config
    batch = args.batch or default_batch
    utils.set_seed(config.seed)
    np.random.seed(config.seed + rank)
    config.local_rank, config.gpus_num, config.group = device.index, world, None
    config.batch_size = batch * world
    config.host_sync_lag = 2
    config.print_interval = 10 ** 9
    config.use_ema_model = getattr(config, 'use_ema_model', False)
    # whole-step capture where the loop allows it (r04): a criterion without host reads and with static shapes (RetinaLoss with
    # SmoothL1); FCOS (positive-only IoU terms) and the SAM loop (prompt draws) stay eager
    is_det = LOOP_MODELS[name][0].endswith('train_detection')
    is_detr = 'detr' in getattr(config, 'network', '')
    # r05: DETR captured whole -- its Hungarian assignment runs on the device (DETRLoss.assign_device, saicv_detr_assign)
    # (the SAM loop's captured step is opt-in, SAICV_SAM_GRAPH=1 in its config: it gains nothing on one GPU -- DESIGN.md section 3k)
    graphed = bool(use_graph and ((not is_det and getattr(config, 'use_step_graph', False)) or
                                  (is_det and not is_detr and getattr(config.train_criterion, 'capturable', False)) or
                                  (is_detr and getattr(config.train_criterion, 'static_form', False))))
    config.use_step_graph = graphed
    model = config.model.to(device)
    optimizer, _ = utils.build_optimizer(config, model)
    scheduler = utils.Scheduler(config, optimizer)
    model, config.ema_model, config.scaler = utils.build_training_mode(config, model)
    data = config.train_collater([config.train_dataset[rank * batch + i] for i in range(batch)])
    data = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}
    iters_per_epoch = max(1, len(config.train_dataset) // config.batch_size)
    logger = logging.getLogger('saicv_bench')
    logger.addHandler(logging.NullHandler())
    logger.propagate = False
    fn = scripts.train_detection if LOOP_MODELS[name][0].endswith('train_detection') else interactive_segmentation_scripts.train_sam_segmentation
    state = {'loss': float('nan')}

    def run(k):
        state['loss'] = fn(DeviceLoader([data] * k, config.batch_size, iters_per_epoch), model, config.train_criterion,
                           optimizer, scheduler, 1, logger, config)
        graphs = getattr(config, '_saicv_step_graphs', None)
        state['step_graph'] = max(graphs.values(), key=lambda g: g.replays) if graphs else None
        state['step_graphs'] = {'captured': sum(g.graph is not None for g in graphs.values()), 'replays': sum(g.replays for g in graphs.values())} if graphs else None

    info = {'config_file': cfg_path, 'loop': loop_name, 'optimizer': config.optimizer[0], 'param_groups': len(optimizer.param_groups)}
    state['graphed'] = graphed
    return run, model, config.scaler, state, info, batch, size


def step_workload(name, args, world, rank, device):
    """SAM image encoder / DETR: a hand-written step (the reference has no encoder-only loop in scope; DETR's loop
    does a host-side Hungarian assignment per step)."""
    import torch
    from simpleaicv_pytorch_training_examples_amd import engine
    batch = args.batch or 8
    size = IMAGE_SIZE[name]
    torch.manual_seed(0)
    g = torch.Generator(device='cpu').manual_seed(1 + rank)
    masks = None
    if name == 'sam_b_encoder':
        # BASELINE.json configs[4]: SAM ViT-B image encoder, 3x1024x1024, trained as in the reference's encoder-distillation
        # setup (MSE against a fixed embedding).  288 GB of HBM hold every block's activations, so the reference's
        # use_gradient_checkpoint=True recompute (sam_b_training/train_config.py:21) is not needed.
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
        model = ViTImageEncoder(image_size=1024, patch_size=16, inplanes=3, embedding_planes=768, block_nums=12, head_nums=12,
                                mlp_ratio=4, out_planes=256, window_size=14, global_attn_indexes=[2, 5, 8, 11],
                                use_gradient_checkpoint=False).to(device)
        crit = lambda out, tgt: torch.nn.functional.mse_loss(out.float(), tgt)
        opt = engine.AdamW(model, [{'params': list(model.parameters()), 'weight_decay': 0.0}], lr=1e-5)
        images = torch.randn(batch, 3, size, size, generator=g).to(device)
        labels = torch.randn(batch, 256, size // 16, size // 16, generator=g).to(device)
        clip = 1.0
    else:
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
        model = detr.resnet50_detr(num_classes=80).to(device)
        loss_mod = DETRLoss(num_classes=80)
        crit = lambda outs, tgt: sum(loss_mod(outs, tgt).values())
        bb = [p for n, p in model.named_parameters() if n.startswith('backbone.')]
        rest = [p for n, p in model.named_parameters() if not n.startswith('backbone.')]
        opt = engine.AdamW(model, [{'params': bb, 'weight_decay': 1e-3, 'lr': 1e-5}, {'params': rest, 'weight_decay': 1e-3}], lr=1e-4)
        canvas = torch.zeros(batch, size, size, 3)
        canvas[:, :800, :, :] = torch.randn(batch, 800, size, 3, generator=g)
        images = canvas.to(device).permute(0, 3, 1, 2)
        masks = torch.ones(batch, size, size, dtype=torch.bool)
        masks[:, :800, :] = False
        masks = masks.to(device)
        labels = -torch.ones(batch, 100, 5)
        labels[:, :10, 0:2] = torch.rand(batch, 10, 2, generator=g) * 0.5 + 0.25
        labels[:, :10, 2:4] = torch.rand(batch, 10, 2, generator=g) * 0.3 + 0.05
        labels[:, :10, 4] = torch.randint(0, 80, (batch, 10), generator=g).float()
        labels = labels.to(device)
        clip = 0.1
    ddp = engine.DistributedDataParallel(model, device_ids=[device.index])
    scaler = engine.GradScaler(device=device)
    ddp.train()
    state = {'loss': float('nan')}

    def one():
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = ddp(images, masks) if masks is not None else ddp(images)
            loss = crit(out, labels)
        scaler.scale(loss).backward()
        ddp.finish_gradient_sync()
        opt.check_finite()          # the reference loop: unscale, clip the global norm, step (tools/scripts.py:1029-1049)
        opt.clip_grad_norm_(clip, scaler.state[2:3])
        opt.step(None, opt.found_inf)
        scaler._found_inf = opt.found_inf
        scaler.update()
        return loss

    def run(k):
        for _ in range(k):
            loss = one()
        state['loss'] = float(loss.detach())

    info = {'loop': 'bench.py step (forward, loss, backward, unscale + clip, fused AdamW)', 'optimizer': 'AdamW'}
    return run, ddp, scaler, state, info, batch, size


def _replay_host_ms(state):
    g = state.get('step_graph') if isinstance(state, dict) else None
    if g is None or not getattr(g, 'replays', 0):
        return None
    return round(g.replay_host_s / g.replays * 1e3, 3)


def measure(name, args, world, rank, device, use_graph, primary):
    """-> result dict of one workload (timed windows, kernel pricing pass)."""
    import torch
    import torch.distributed as dist
    from simpleaicv_pytorch_training_examples_amd import ops
    if name in LOOP_MODELS:
        run, model, scaler, state, info, batch, size = loop_workload(name, args, world, rank, device, use_graph)
        use_graph = state['graphed']
    elif name in CONFIG_DIR:
        run, model, scaler, state, info, batch, size = classification_workload(name, args, world, rank, device, use_graph)
    else:
        use_graph = False
        run, model, scaler, state, info, batch, size = step_workload(name, args, world, rank, device)
    ops.KernelTimer.enabled = False

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: eager iterations first (the step graph is captured after 3 of them), then replays
    run(max(args.warmup, 5 if use_graph else 1))
    windows, host = [], []
    budget_windows = args.max_windows if primary else max(1, args.max_windows // 3)
    while True:
        fence()
        t0 = time.perf_counter()
        run(args.steps)
        th = time.perf_counter() - t0
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        windows.append(dt)
        host.append(th)
        if len(windows) >= budget_windows or sum(windows) >= (args.min_gpu_seconds if primary else args.min_gpu_seconds / 2):
            break
    med = statistics.median(windows)
    ms = med / args.steps * 1e3
    value = batch * world * args.steps / med
    arena = model.arena if hasattr(model, 'arena') else None
    res = {
        'value': round(value, 1), 'ms_per_step': round(ms, 3),
        'windows_ms_per_step': [round(w / args.steps * 1e3, 3) for w in windows],
        'config': {'workload': f'{name} 3x{size}x{size} synthetic training step (fwd+loss+bwd+all-reduce+optimizer), per-GPU batch {batch}',
                   'model': name, 'global_batch': batch * world, 'per_gpu_batch': batch, 'parallelism': f'dp{world}',
                   'final_loss': round(float(state['loss']), 4), 'loss_scale': scaler.get_scale() if scaler is not None else None,
                   'step_graph': bool(use_graph), 'step_graphs': state.get('step_graphs') if isinstance(state, dict) else None,
                   'bn_statistics': 'fp32 atomics into pooled rows (SAICV_BN_INLINE=1)' if ops.BN_INLINE else 'fixed-order partial rows',
                   'host_ms_per_step': round(statistics.median(host) / args.steps * 1e3, 3),
                   # host time spent ISSUING a step when it is a graph replay (input copies, hyper-parameter refresh,
                   # hipGraphLaunch); host_ms_per_step above also contains the loop's lagged read of the loss, i.e. waiting
                   'host_enqueue_ms_per_step': _replay_host_ms(state), **info},
        'model_mfma_frac': round((LOOP_MODELS[name][3] if name in LOOP_MODELS else TRAIN_GFLOP_PER_IMG.get(name, 0)) * value / world / 1e3 / PEAK_BF16_TFLOPS, 4),
        'rccl_ranks': dist.get_world_size() if world > 1 else 1,
        'gradient_allreduce': ('saicv_comm (library RCCL communicator)' if getattr(model, 'comm', None) is not None
                               else 'torch.distributed') if world > 1 else None,
        # True: the bucketed all-reduces run on the communication stream under the rest of backward (captured step);
        # False: serialised on the compute stream (eager launches)
        'overlap': bool(use_graph) if world > 1 else None,
        'allreduce_bytes_per_step': int(sum(b['end'] - b['start'] for b in model.buckets) * 4) if (world > 1 and hasattr(model, 'buckets')) else 0,
        'gradient_bytes': int(arena.total * 4) if arena is not None else None,
    }
    # ---- socket power and shader clock while the same steps replay (rocm-smi polled from a thread for ~1.5 s, AFTER the timed windows).
    # r05 finding: the GEMM kernels of these steps run AT the 1 400 W socket limit on random operands and the shader clock gives way
    # (2.0-2.2 of 2.4 GHz): `power` says how close the whole step sits to that limit.
    if world == 1 and not getattr(args, 'no_power', False):
        res['power'] = sample_power(lambda: run(args.steps), fence)
    # ---- price the dominant kernel: an eager pass of the same steps with HIP events on the launch stream
    if not args.no_kernel_timer:
        cfg_graph = name in CONFIG_DIR and use_graph
        ops.KernelTimer.only = {'igemm_nt'} if args.dominant_only else None
        ops.KernelTimer.records = []
        k = min(args.steps, 5)
        if cfg_graph:
            _set_graph(name, False)
        run(1)
        fence()
        ops.KernelTimer.enabled = True
        run(k)
        fence()
        ops.KernelTimer.enabled = False
        if cfg_graph:
            _set_graph(name, True)
        summ = ops.KernelTimer.summary()
        if 'igemm_nt' in summ:
            kk = summ['igemm_nt']
            achieved = kk['flops'] / (kk['ms'] * 1e-3) / 1e12
            traffic, traffic_file = pmc_traffic(name, 'igemm_nt')
            res['roofline'] = {'kernel': 'igemm_nt_kernel (implicit-GEMM conv / linear, fwd + dgrad)', 'bound': 'mfma',
                               'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                               'traffic_source': (f'{traffic_file} (separate rocprofv3 --pmc passes of this command, not this run)'
                                                  if traffic is not None else None),
                               # the same traffic read from the L2's fabric REQUEST counters (TCC_EA0_RDREQ / WRREQ x 64 B, 32-byte reads at
                               # 32 B; scripts/make_tcc_traffic.py): FETCH_SIZE's x 2 rule for gfx950 assumes 128-byte reads and over-counts
                               # the 64-byte K-slice requests of the LDS-DMA loads, this reading under-counts true 128-byte requests --
                               # the two bracket the real figure
                               'traffic_fabric_requests': tcc_traffic(name, 'igemm_nt', kk['bytes'] / max(kk['calls'], 1)),
                               'launches': kk['calls'], 'avg_launch_us': round(kk['ms'] * 1e3 / kk['calls'], 2),
                               # what the SHAPES allow: sum over the launches of max(flops / MFMA peak, algorithmic bytes /
                               # HBM peak) -- many ResNet-50 launches (K <= 256, and the data gradients that also carry a
                               # BatchNorm-backward reduction in their epilogue) are HBM-bound at these peaks
                               'shape_bound': {'lower_bound_ms': round(kk['bound_ms'] / k, 3), 'measured_ms': round(kk['ms'] / k, 3),
                                               'frac': round(kk['bound_ms'] / kk['ms'], 4) if kk['ms'] else None,
                                               'algorithmic_GB': round(kk['bytes'] / k / 1e9, 2)},
                               'measured_in': f'{k} eager steps after the timed windows (HIP events on the launch stream; '
                                              'events cannot bracket kernels inside a replayed hipGraph)'}
            top = sorted(summ.items(), key=lambda tv: -tv[1]['ms'])
            res['kernel_breakdown_ms_per_step'] = {t: round(v['ms'] / k, 3) for t, v in top}
            res['kernel_breakdown_note'] = ('HIP events around every launch of the instrumented families in the eager pricing pass '
                                            '(they cover the GEMM, normalisation and attention kernels; pooling, loss, optimizer and '
                                            'tensor glue are the rest of ms_per_step)')
            for t, v in top:
                if t == 'igemm_nt':
                    continue            # priced against the MFMA peak in `roofline` (its bytes are in roofline.shape_bound)
                fam = {'kernel': FAMILY_KERNELS.get(t, t), 'ms_per_step': round(v['ms'] / k, 3), 'launches_per_step': round(v['calls'] / k, 1)}
                hbm_bound = v['bytes'] > 0 and v['bytes'] / 8e12 >= v['flops'] / (PEAK_BF16_TFLOPS * 1e12)
                if hbm_bound:
                    # memory-bound families: algorithmic bytes (tensors read + written once) / measured time against HBM3E
                    fam.update({'bound': 'hbm', 'algorithmic_GB_per_step': round(v['bytes'] / k / 1e9, 2),
                                'GB/s': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1),
                                'frac_of_8TBps': round(v['bytes'] / (v['ms'] * 1e-3) / 8e12, 4)})
                    pm, _ = pmc_traffic(name, {'bn_act_bwd': 'bn_bwd_apply'}.get(t, t.replace('_fwd', '').replace('_bwd', '') if t.startswith('layernorm') else t))
                    if pm is not None:
                        fam['pmc_bytes_per_launch'] = pm
                    res.setdefault('hbm_kernels', {})[t] = fam
                elif v['flops'] > 0:
                    fam.update({'bound': 'mfma', 'TFLOP/s': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1),
                                'frac_of_mfma_peak': round(v['flops'] / (v['ms'] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)})
                    res.setdefault('mfma_kernels', {})[t] = fam
    return res


def sample_power(run_steps, fence, seconds=1.5):
    """-> {'socket_w', 'sclk_mhz', 'samples', 'limit_w'} averaged over `seconds` of replayed steps, or None when rocm-smi is not usable."""
    import re
    import shutil
    import threading
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    if not os.path.exists(exe):
        return None
    stop, got = threading.Event(), []

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run([exe, '--showpower', '--showclocks', '--json'], capture_output=True, text=True, timeout=5).stdout
                w = re.search(r'Power \(W\)": "([0-9.]+)', out)
                c = re.search(r'sclk clock speed:": "\((\d+)Mhz', out)
                if w and c:
                    got.append((float(w.group(1)), int(c.group(1))))
            except Exception:      # noqa: BLE001  (a missing / hanging tool must not fail the bench)
                return
            time.sleep(0.1)
    th = threading.Thread(target=poll, daemon=True)
    fence()
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        run_steps()
    fence()
    stop.set()
    th.join(timeout=6)
    got = got[1:] if len(got) > 2 else got          # the first sample may predate the load
    if not got:
        return None
    return {'socket_w': round(sum(g[0] for g in got) / len(got), 1), 'sclk_mhz': round(sum(g[1] for g in got) / len(got)),
            'samples': len(got), 'limit_w': 1400, 'max_sclk_mhz': 2400,
            'note': 'rocm-smi --showpower --showclocks polled while the timed workload replays (after the timed windows)'}


def want_step_graph(eager, force_graph, world, env):
    """Whether the training step runs as one replayed hipGraph.  One GPU: yes unless --eager.  Several ranks (the driver's
    `--gpus N`): ALSO yes by default -- inside a captured step the bucketed all-reduces run on the library's communication
    stream behind event edges and overlap the rest of backward (csrc/comm.hip), which is what nn.parallel.DistributedDataParallel
    gives the reference loop (reference tools/utils.py:193-197); launched eagerly they are serialised on the compute stream
    (profiles/r02_ddp_eager_path.md).  SAICV_STEP_GRAPH=0 / --eager select the eager path, and it is the fallback when the
    capture raises or (watchdog) the first replays do not come back."""
    if eager or env == '0':
        return False
    return True if world == 1 else (force_graph or env in (None, '', '1'))


class WatchdogVote:
    """N > 1 only.  A captured N-rank step whose RCCL collective never completes inside a replay cannot raise, so a timer guards
    warm-up + capture + the first timed windows.  The DECISION is collective: a rank that times out locally while
    another one just finished must not re-execute alone -- the ranks would then disagree on graph vs eager and on the RCCL
    communicator state, which is the hang the watchdog exists to prevent.  Protocol over the job's key-value store (the
    torch.distributed default store; host-side TCP, usable while the main thread sits in a GPU wait):
        every rank:   set  wd/<epoch>/done/<rank>   when its guarded section finished;
        rank 0:       at the deadline -- or as soon as every done key exists -- publishes  wd/<epoch>/decision = "ok" | "reexec";
        every rank:   blocks (in its watchdog thread) on that ONE key and acts on it: "reexec" -> the whole job re-executes
                      itself with --eager (same PIDs: the launcher keeps its children; fresh HIP / RCCL state).
    `act` is injectable so that tests can drive the protocol with threads and a HashStore."""

    def __init__(self, store, rank, world, seconds, epoch, act=None, poll=0.2):
        import threading
        self.store, self.rank, self.world, self.seconds, self.epoch = store, rank, world, seconds, epoch
        self.act = act or self._reexec
        self.poll = poll
        self.decision = None
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _key(self, what):
        return f'wd/{self.epoch}/{what}'

    def finished(self):
        """The guarded section of THIS rank is over (called from the main thread)."""
        self.store.set(self._key(f'done/{self.rank}'), b'1')

    def _all_done(self):
        keys = [self._key(f'done/{r}') for r in range(self.world)]
        try:
            return bool(self.store.check(keys))
        except Exception:       # noqa: BLE001 -- a store that cannot answer counts as "not done"
            return False

    def _run(self):
        deadline = time.time() + self.seconds
        if self.rank == 0:
            verdict = b'reexec'
            while time.time() < deadline:
                if self._all_done():
                    verdict = b'ok'
                    break
                time.sleep(self.poll)
            else:
                verdict = b'ok' if self._all_done() else b'reexec'
            self.store.set(self._key('decision'), verdict)
        # every rank (rank 0 included) reads the one published decision; other ranks wait as long as rank 0 may take
        while True:
            try:
                if self.store.check([self._key('decision')]):
                    break
            except Exception:   # noqa: BLE001
                pass
            if time.time() > deadline + 60:
                # rank 0 itself is gone or its store unreachable: nothing collective is possible any more
                self.decision = 'reexec'
                return self.act(self.seconds, 'no decision from rank 0')
            time.sleep(self.poll)
        self.decision = self.store.get(self._key('decision')).decode()
        if self.decision == 'reexec':
            self.act(self.seconds, 'collective decision')

    @staticmethod
    def _reexec(seconds, why):
        sys.stderr.write(f'[bench] captured N-rank step did not finish within {seconds:.0f} s on every rank ({why}): '
                         're-executing with --eager\n')
        sys.stderr.flush()
        os.environ['SAICV_BENCH_REEXEC'] = '1'
        argv = [a for a in sys.argv if a != '--graph'] + ['--eager']
        os.execv(sys.executable, [sys.executable] + argv)


def _arm_last_resort(seconds, world):
    """After a watchdog re-execution (or with SAICV_BENCH_HARD_LIMIT_S set): if even the eager run does not finish, rank 0
    prints a diagnosable JSON line -- value null, the reason -- and every rank exits non-zero instead of hanging the driver."""
    import threading

    def run():
        time.sleep(seconds)
        if int(os.environ.get('RANK', '0')) == 0:
            print(json.dumps({'metric': 'training images/sec/node', 'value': None, 'unit': 'images/s', 'n_gpus': world,
                              'higher_is_better': True, 'error': f'bench.py did not finish within {seconds:.0f} s after falling back '
                              'to eager launches (a rank is stuck in a collective or died): no number is reported',
                              'graph_fallback': os.environ.get('SAICV_BENCH_REEXEC') == '1'}), flush=True)
        os._exit(3)

    threading.Thread(target=run, daemon=True).start()


_CONFIGS = {}


def _set_graph(name, on):
    cfg = _CONFIGS.get(name)
    if cfg is not None:
        cfg.use_step_graph = on


def pmc_traffic(model, kernel):
    """(HBM bytes per launch of `kernel` in `model`'s step, file) from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over this same command, corrected as MI355X_MICROARCH.md prescribes); (None, None) when no summary holds it.
    Counters cannot be collected inside the timed run itself."""
    for f in PMC_FILES.get(model, []):
        try:
            return json.load(open(os.path.join(ROOT, f)))['kernels'][kernel]['bytes_per_launch'], f
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def tcc_traffic(model, kernel, algorithmic_bytes_per_launch):
    """{'bytes_per_launch', 'ratio_to_algorithmic', 'source'} from the committed TCC request-counter summary of this command, or None."""
    f = 'profiles/r06_tcc_traffic' + ('' if model == 'resnet50' else '_' + model) + '.json'
    try:
        b = json.load(open(os.path.join(ROOT, f)))['kernels'][kernel]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None
    return {'bytes_per_launch': b, 'ratio_to_algorithmic': round(b / algorithmic_bytes_per_launch, 3) if algorithmic_bytes_per_launch else None,
            'source': f'{f} (a rocprofv3 --pmc pass of its own over this command, not this run)'}


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(model_name):
    """fp32 train steps (fwd + loss + bwd + SGD) of ResNet-50 at batch 16 on the host cores: the reference's own modules
    and torch.optim.SGD when /root/reference is importable (the build container), else the CPU oracle restatement
    (the GPU box has no /root/reference).  Threads = physical cores (capped at 64: a batch-16 step does not scale
    further and oversubscribed SMT threads made round 1's figure unstable); median of >= 5 steps after one warm-up."""
    import torch
    if model_name != 'resnet50':
        return None
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count() or 8
    except ImportError:
        phys = max(1, (os.cpu_count() or 8) // 2)
    threads = int(os.environ.get('SAICV_CPU_BASELINE_THREADS', min(phys, 64)))
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    b = 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, 224, 224, 3, generator=g).permute(0, 3, 1, 2)
    y = torch.randint(0, 1000, (b,), generator=g)
    if os.path.isdir('/root/reference/SimpleAICV') and not os.environ.get('SAICV_CPU_BASELINE_PORT'):
        # the reference's own modules, in a child process whose import root is /root/reference (this process has
        # `SimpleAICV` aliased to the MI355X package)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-reference-child', str(threads)],
                               capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
            torch.set_num_threads(old)
            return json.loads(line)
        except Exception as e:      # noqa: BLE001 -- any problem falls back to the oracle port, and says so
            print(f'[bench] reference modules not usable for the CPU baseline ({e}); using the oracle port', file=sys.stderr)
    from oracle import torch_oracle as O
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    torch.manual_seed(0)
    m = backbones.resnet50(num_classes=1000)
    sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in m.state_dict().items()}
    pnames = [n for n, _ in m.named_parameters()]
    wd = {n: (1e-4 if sd[n].ndim > 1 else 0.0) for n in pnames}
    bufs = {}

    def step():
        nonlocal bufs
        _, _, grads = O.loss_and_grads(lambda lv, inp: O.resnet_forward('resnet50', lv, inp, True), sd, pnames, x,
                                       loss_fn=O.ce_loss, label=y)
        params, bufs = O.sgd_momentum_step({n: sd[n] for n in pnames}, grads, bufs, 0.1, 0.9, wd)
        sd.update(params)
    res = _time_cpu_steps(step, b, threads, 'port')
    torch.set_num_threads(old)
    return res


def _time_cpu_steps(step, b, threads, kind):
    step()
    times = []
    t_all = time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t_all < 12 and len(times) < 15):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    what = ('the reference\'s own ResNet-50 / CELoss modules + torch.optim.SGD' if kind == 'reference'
            else 'the CPU oracle restatement (oracle/torch_oracle.py; /root/reference is not present on this box)')
    return {'value': round(b / dt, 2), 'unit': 'images/s', 'cores': threads, 'kind': kind,
            'sample': f'median of {len(times)} fp32 train steps (fwd+loss+bwd+SGD) of {what} at batch {b}, 224x224, '
                      f'{threads} threads of {os.cpu_count()} logical CPUs'}


def cpu_reference_child(threads):
    """Runs in a child process: import root = /root/reference, nothing of this repository imported."""
    import torch
    sys.path.insert(0, '/root/reference')
    from SimpleAICV.classification.backbones.resnet import resnet50
    from SimpleAICV.classification.losses import CELoss
    torch.set_num_threads(threads)
    b = 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, 224, 224, 3, generator=g).permute(0, 3, 1, 2)
    y = torch.randint(0, 1000, (b,), generator=g)
    torch.manual_seed(0)
    m = resnet50(num_classes=1000)
    m.train()
    crit = CELoss()
    decay = [p for p in m.parameters() if p.ndim > 1]
    plain = [p for p in m.parameters() if p.ndim <= 1]
    opt = torch.optim.SGD([{'params': decay, 'weight_decay': 1e-4}, {'params': plain, 'weight_decay': 0.0}], lr=0.1, momentum=0.9)

    def step():
        opt.zero_grad()
        crit(m(x), y).backward()
        opt.step()
    print(json.dumps(_time_cpu_steps(step, b, threads, 'reference')), flush=True)


# ------------------------------------------------------------------------------------------ worker
def worker(args):
    # BatchNorm statistics: the entry scripts' set_seed() selects the fixed-order (bit-reproducible) sums, as the reference's
    # set_seed asks for deterministic kernels; the benchmark opts into the atomically accumulated ones (0.35-0.4 ms per
    # ResNet-50 step) unless --deterministic, and says so in its JSON line (config.bn_statistics)
    if not args.deterministic:
        os.environ.setdefault('SAICV_BN_INLINE', '1')
        os.environ.setdefault('SAICV_DETERMINISTIC', '0')
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks for --gpus N '
                         '(python bench.py --gpus N starts them itself)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs MI355X GPUs (the HIP path has no CPU fallback)')
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f'bench.py: rank {rank} wants cuda:{local_rank} but only {torch.cuda.device_count()} GPUs are visible')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', init_method='env://', device_id=device)
        assert dist.get_world_size() == world and dist.get_backend() == 'nccl'
        probe = torch.ones(1, device=device)
        dist.all_reduce(probe)                         # RCCL really connects `world` ranks before anything is timed
        assert int(probe) == world, f'RCCL all-reduce over {world} ranks returned {float(probe)}'
    want_graph = want_step_graph(args.eager, args.graph, world, os.environ.get('SAICV_STEP_GRAPH'))
    reexeced = os.environ.get('SAICV_BENCH_REEXEC') == '1'
    wd_seconds = float(os.environ.get('SAICV_BENCH_WATCHDOG_S', '300'))
    guard = world > 1 and want_graph and not reexeced

    def new_watchdog(epoch):
        from torch.distributed import distributed_c10d as c10d
        return WatchdogVote(c10d._get_default_store(), rank, world, wd_seconds, epoch)

    if world > 1 and (reexeced or os.environ.get('SAICV_BENCH_HARD_LIMIT_S')):
        _arm_last_resort(float(os.environ.get('SAICV_BENCH_HARD_LIMIT_S', str(3 * wd_seconds))), world)
    watchdog = new_watchdog(0) if guard else None
    if os.environ.get('SAICV_BENCH_INJECT_STALL') == str(rank) and not reexeced:
        time.sleep(wd_seconds + 30)         # test hook: this rank never reaches its first collective in time

    def guarded(name, primary):
        try:
            return measure(name, args, world, rank, device, want_graph, primary)
        except Exception as e:      # noqa: BLE001
            if not want_graph or name not in CONFIG_DIR or name in LOOP_MODELS or world > 1:
                # several ranks: a rank-local fallback would split the job (graph here, eager there); the failure is fatal
                # and the launcher reports it
                raise
            print(f'[bench] step graph failed for {name} ({type(e).__name__}: {e}); falling back to eager launches', file=sys.stderr)
            torch.cuda.synchronize()
            return measure(name, args, world, rank, device, False, primary)

    primary = guarded(args.model, True)
    if watchdog is not None:
        watchdog.finished()             # this rank's captured step replayed and was timed; rank 0 decides for everybody
    secondary = None
    if args.model == 'resnet50' and not args.no_secondary:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        if guard:                       # the secondary workload captures its own step: same vote, next epoch
            watchdog = new_watchdog(1)
        secondary = guarded('vit_base_patch16', False)
        if guard:
            watchdog.finished()
    sam = None
    if args.model == 'resnet50' and not args.no_secondary and not args.no_sam and world == 1:
        # BASELINE.json configs[4] on the driver line: SAM ViT-B image encoder, 3 x 1024 x 1024, the reference's per-GPU batch of
        # 20 (sam_b_training/train_config.py:221), ONE timed window of 5 steps (~0.6 s) + the kernel pricing pass
        import argparse as _ap
        import gc
        for cfg in _CONFIGS.values():           # drop the captured step graphs (and their memory pools) of the two classification runs
            for attr in ('_saicv_step_graphs', 'model', 'ema_model'):
                if hasattr(cfg, attr):
                    setattr(cfg, attr, None)
        _CONFIGS.clear()
        gc.collect()
        torch.cuda.empty_cache()
        a2 = _ap.Namespace(**vars(args))
        a2.batch, a2.steps, a2.warmup, a2.max_windows, a2.min_gpu_seconds = 20, 5, 2, 1, 0.0
        try:
            sam = measure('sam_b_encoder', a2, world, rank, device, False, False)
            sam['steps'], sam['warmup'] = a2.steps, a2.warmup
        except Exception as e:      # noqa: BLE001 -- the headline must not die with the third object
            sam = {'error': f'{type(e).__name__}: {e}'}
    detr = None
    if args.model == 'resnet50' and not args.no_secondary and not args.no_sam and world == 1:
        # BASELINE.json configs[3] on the driver line (r05): the reference's DETR-R50 config through tools.scripts.train_detection,
        # per-GPU batch 8, the whole step captured (Hungarian assignment on the device); ONE window of 5 steps
        import argparse as _ap
        import gc
        for cfg in _CONFIGS.values():
            for attr in ('_saicv_step_graphs', 'model', 'ema_model'):
                if hasattr(cfg, attr):
                    setattr(cfg, attr, None)
        _CONFIGS.clear()
        gc.collect()
        torch.cuda.empty_cache()
        a3 = _ap.Namespace(**vars(args))
        a3.batch, a3.steps, a3.warmup, a3.max_windows, a3.min_gpu_seconds, a3.no_kernel_timer = 8, 5, 5, 1, 0.0, False      # (r06: priced like the others)
        try:
            detr = measure('resnet50_detr_config', a3, world, rank, device, want_step_graph(args.eager, args.graph, world, os.environ.get('SAICV_STEP_GRAPH')), False)
            detr['steps'], detr['warmup'] = a3.steps, a3.warmup
        except Exception as e:      # noqa: BLE001
            detr = {'error': f'{type(e).__name__}: {e}'}

    sam_full = None
    if args.model == 'resnet50' and not args.no_secondary and not args.no_sam and world == 1:
        # BASELINE.json configs[4] as the reference trains it (r06): the FULL SAM step -- image encoder + 1 + decoder_iters prompt /
        # decoder passes + SAMLoss + AdamW -- through tools.interactive_segmentation_scripts.train_sam_segmentation and the reference's
        # config, per-GPU batch 8 (eager launches unless SAICV_SAM_GRAPH=1: then one graph per drawn prompt combination, the 12 warm-up
        # steps capture both of the config's)
        import argparse as _ap
        import gc
        for cfg in _CONFIGS.values():
            for attr in ('_saicv_step_graphs', 'model', 'ema_model'):
                if hasattr(cfg, attr):
                    setattr(cfg, attr, None)
        _CONFIGS.clear()
        gc.collect()
        torch.cuda.empty_cache()
        a4 = _ap.Namespace(**vars(args))
        a4.batch, a4.steps, a4.warmup, a4.max_windows, a4.min_gpu_seconds, a4.no_kernel_timer = 8, 5, 12, 1, 0.0, False
        try:
            sam_full = measure('sam_b', a4, world, rank, device, want_step_graph(args.eager, args.graph, world, os.environ.get('SAICV_STEP_GRAPH')), False)
            sam_full['steps'], sam_full['warmup'] = a4.steps, a4.warmup
        except Exception as e:      # noqa: BLE001
            sam_full = {'error': f'{type(e).__name__}: {e}'}

    if rank == 0:
        out = {'metric': 'training images/sec/node', 'value': primary['value'], 'unit': 'images/s', 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': primary['ms_per_step'], 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic'}
        out.update({k: v for k, v in primary.items() if k not in ('value', 'ms_per_step')})
        if reexeced:
            out['graph_fallback'] = ('the captured N-rank step did not finish within the watchdog limit on every rank; the job '
                                     're-executed itself with --eager (collective decision over the process-group store)')
        out['timing'] = (f'median of {len(primary["windows_ms_per_step"])} windows of exactly {args.steps} steps, each bracketed by '
                         'barrier + synchronize (max over ranks)')
        if secondary is not None:
            out['secondary'] = {'metric': 'training images/sec/node', 'unit': 'images/s', **secondary}
        if sam is not None:
            out['sam_b_encoder'] = {'metric': 'training images/sec/node', 'unit': 'images/s', **sam}
        if detr is not None:
            out['resnet50_detr_config'] = {'metric': 'training images/sec/node', 'unit': 'images/s', **detr}
        if sam_full is not None:
            out['sam_b'] = {'metric': 'training images/sec/node', 'unit': 'images/s', **sam_full}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model)
        line = json.dumps(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio, which would otherwise be flushed at exit, AFTER this line:
        # flush it first so that the JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(line, flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-reference-child':
        return cpu_reference_child(int(sys.argv[2]))
    args = parse()
    if 'WORLD_SIZE' not in os.environ and (args.gpus or 1) > 1:
        sys.exit(spawn(args))
    if args.gpus is None:
        args.gpus = int(os.environ.get('WORLD_SIZE', '1'))
    worker(args)


if __name__ == '__main__':
    main()
