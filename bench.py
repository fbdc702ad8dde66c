"""Training-throughput bench of the SimpleAICV DDP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model resnet50|vit_base_patch16]
                    [--batch B] [--no-cpu-baseline]

A step = forward + loss + backward + gradient all-reduce + optimizer step of one per-GPU
batch of synthetic ImageNet-shape data (BASELINE.json configs[1]: ResNet-50, 224x224, bf16,
per-GPU batch 256; weak scaling over N GPUs, one process per GPU over RCCL).  Prints ONE JSON
line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (implicit-GEMM conv, MFMA-bound) priced from HIP events
                  recorded around every one of its launches inside the timed region;
  cpu_baseline -- the CPU oracle (fp32 restatement of the reference, oracle/torch_oracle.py)
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md
# SURVEY.md section 8(d); DETR: 189.1 GFLOP fwd at 800x1344, scaled to the 1333x1333 canvas the collater pads to
TRAIN_GFLOP_PER_IMG = {'resnet50': 24.54, 'vit_base_patch16': 105.38, 'sam_b_encoder': 3 * 972.1,
                       'resnet50_detr': 3 * 189.1 * (1333 * 1333) / (800 * 1344)}
IMAGE_SIZE = {'sam_b_encoder': 1024, 'resnet50_detr': 1333}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over this same command, corrected as MI355X_MICROARCH.md prescribes); None when the
    summary is absent.  Counters cannot be collected inside the timed run itself."""
    path = os.path.join(ROOT, 'profiles', 'r01e_pmc_hbm_traffic.json')
    try:
        return json.load(open(path))['kernels'][kernel]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='resnet50')
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--kernel-breakdown', action='store_true',
                    help='bracket every kernel family with HIP events (adds host overhead; default: only the dominant kernel)')
    return ap.parse_args()


def build(model_name, device):
    from simpleaicv_pytorch_training_examples_amd import engine
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones, losses
    torch.manual_seed(0)
    if model_name == 'resnet50':
        model = backbones.resnet50(num_classes=1000).to(device)
        crit = losses.CELoss()
        soft = False
    elif model_name == 'vit_base_patch16':
        model = backbones.vit_base_patch16(image_size=224, drop_path_prob=0.1, global_pool=True,
                                           num_classes=1000).to(device)
        crit = losses.OneHotLabelCELoss()
        soft = True
    elif model_name == 'sam_b_encoder':
        # BASELINE.json configs[4]: SAM ViT-B image encoder, 3x1024x1024.  Trained as in the reference's
        # encoder-distillation setup (13.0.encoder_distill_training): MSE against a fixed embedding.
        # 288 GB of HBM hold every block's activations at per-GPU batch 20, so the reference's
        # use_gradient_checkpoint=True recompute (train_config.py:21) is not needed.
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.interactive_segmentation.models.segment_anything.image_encoder import ViTImageEncoder
        model = ViTImageEncoder(image_size=1024, patch_size=16, inplanes=3, embedding_planes=768, block_nums=12,
                                head_nums=12, mlp_ratio=4, out_planes=256, window_size=14,
                                global_attn_indexes=[2, 5, 8, 11], use_gradient_checkpoint=False).to(device)
        crit = lambda out, tgt: torch.nn.functional.mse_loss(out.float(), tgt)
        soft = 'embedding'
    elif model_name == 'resnet50_detr':
        # BASELINE.json configs[3]: DETR-ResNet50, COCO-shape 3x800x1333 images on the square canvas
        # DETRDetectionCollater(resize_type='retina_style') pads to (1333 x 1333), padding mask included
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.models import detr
        from simpleaicv_pytorch_training_examples_amd.SimpleAICV.detection.losses import DETRLoss
        model = detr.resnet50_detr(num_classes=80).to(device)
        loss_mod = DETRLoss(num_classes=80)
        crit = lambda outs, tgt: sum(loss_mod(outs, tgt).values())
        soft = 'detr'
    else:
        raise SystemExit(f'unknown model {model_name}')
    return model, crit, soft, engine


def make_optimizer(model_name, model, engine):
    """Optimizer settings of the reference configs: ResNet-50 SGD lr 0.1
    momentum 0.9 wd 1e-4 with 1-d parameters at wd 0 (imagenet/resnet50/train_config.py:71-91);
    ViT-B AdamW lr 5e-4 wd 0.05 (vit_base_patch16.../train_config.py:93-124)."""
    decay = [p for p in model.parameters() if p.ndim > 1]
    no_decay = [p for p in model.parameters() if p.ndim <= 1]
    if model_name == 'resnet50_detr':     # res50_detr_yoloresize1024/train_config.py: AdamW lr 1e-4, wd 1e-3, backbone lr 1e-5
        bb = [p for n, p in model.named_parameters() if n.startswith('backbone.')]
        rest = [p for n, p in model.named_parameters() if not n.startswith('backbone.')]
        return engine.AdamW(model, [{'params': bb, 'weight_decay': 1e-3, 'lr': 1e-5}, {'params': rest, 'weight_decay': 1e-3}],
                            lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    if model_name == 'sam_b_encoder':     # sam_b_training/train_config.py: AdamW lr 1e-5, no weight decay
        return engine.AdamW(model, [{'params': list(model.parameters()), 'weight_decay': 0.0}], lr=1e-5,
                            betas=(0.9, 0.999), eps=1e-8)
    if model_name == 'resnet50':
        return engine.SGD(model, [{'params': decay, 'weight_decay': 1e-4}, {'params': no_decay, 'weight_decay': 0.0}],
                          lr=0.1, momentum=0.9)
    return engine.AdamW(model, [{'params': decay, 'weight_decay': 0.05}, {'params': no_decay, 'weight_decay': 0.0}],
                        lr=5e-4, betas=(0.9, 0.999), eps=1e-8)


def cpu_baseline(model_name):
    """CPU oracle train step (fp32) on the host cores: bounded sample of the same workload."""
    from oracle import torch_oracle as O
    from simpleaicv_pytorch_training_examples_amd.SimpleAICV.classification import backbones
    if model_name != 'resnet50':
        return None
    threads = torch.get_num_threads()
    torch.manual_seed(0)
    m = backbones.resnet50(num_classes=1000)
    sd = {k: v.detach().clone(memory_format=torch.contiguous_format) for k, v in m.state_dict().items()}
    pnames = [n for n, _ in m.named_parameters()]
    b = 16
    x = torch.randn(b, 3, 224, 224)
    y = torch.randint(0, 1000, (b,))
    bufs = {}
    wd = {n: (1e-4 if sd[n].ndim > 1 else 0.0) for n in pnames}

    def step():
        nonlocal sd, bufs
        _, _, grads = O.loss_and_grads(lambda lv, inp: O.resnet_forward('resnet50', lv, inp, True), sd, pnames, x,
                                       loss_fn=O.ce_loss, label=y)
        params = {n: sd[n] for n in pnames}
        params, bufs = O.sgd_momentum_step(params, grads, bufs, 0.1, 0.9, wd)
        sd.update(params)

    step()
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10 and n < 12):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {'value': round(b / dt, 2), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} fp32 train steps (fwd+loss+bwd+SGD) of the CPU oracle ResNet-50 at batch {b}, 224x224'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs MI355X GPUs (the HIP path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', init_method='env://', device_id=device)
    from simpleaicv_pytorch_training_examples_amd import ops

    model, crit, soft, engine = build(args.model, device)
    opt = make_optimizer(args.model, model, engine)
    ddp = engine.DistributedDataParallel(model, device_ids=[local_rank])
    scaler = engine.GradScaler(device=device)

    g = torch.Generator(device='cpu').manual_seed(1 + rank)
    # NCHW-shaped, NHWC-strided fp32 batch, as the reference collater delivers it
    size = IMAGE_SIZE.get(args.model, 224)
    masks = None
    if soft == 'embedding':       # SAMBatchCollater stacks per-sample CHW tensors: true NCHW input
        images = torch.randn(args.batch, 3, size, size, generator=g).to(device)
        labels = torch.randn(args.batch, 256, size // 16, size // 16, generator=g).to(device)
    elif soft == 'detr':          # 800 x 1333 image at the top-left of the canvas, 10 boxes per image
        canvas = torch.zeros(args.batch, size, size, 3)
        canvas[:, :800, :, :] = torch.randn(args.batch, 800, size, 3, generator=g)
        images = canvas.to(device).permute(0, 3, 1, 2)
        masks = torch.ones(args.batch, size, size, dtype=torch.bool)
        masks[:, :800, :] = False
        masks = masks.to(device)
        labels = -torch.ones(args.batch, 100, 5)
        labels[:, :10, 0:2] = torch.rand(args.batch, 10, 2, generator=g) * 0.5 + 0.25
        labels[:, :10, 2:4] = torch.rand(args.batch, 10, 2, generator=g) * 0.3 + 0.05
        labels[:, :10, 4] = torch.randint(0, 80, (args.batch, 10), generator=g).float()
        labels = labels.to(device)
    else:
        images = torch.randn(args.batch, size, size, 3, generator=g).to(device).permute(0, 3, 1, 2)
    if soft in ('embedding', 'detr'):
        pass
    elif soft:
        labels = torch.softmax(torch.randn(args.batch, 1000, generator=g) * 4, -1).to(device)
    else:
        labels = torch.randint(0, 1000, (args.batch,), generator=g).to(device)

    clip = {'resnet50_detr': 0.1, 'sam_b_encoder': 1.0}.get(args.model, 0.0)     # clip_max_norm of the reference configs

    def step():
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            out = ddp(images, masks) if masks is not None else ddp(images)
            loss = crit(out, labels)
        scaler.scale(loss).backward()
        ddp.finish_gradient_sync()
        if clip > 0:        # the reference loop: unscale, clip the global norm, step (tools/scripts.py:1029-1049)
            opt.check_finite()
            opt.clip_grad_norm_(clip, scaler.state[2:3])
            opt.step(None, opt.found_inf)
            scaler._found_inf = opt.found_inf
        else:
            scaler.step(opt)
        scaler.update()
        return loss

    ddp.train()
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.KernelTimer.enabled = not args.no_kernel_timer
    ops.KernelTimer.only = None if args.kernel_breakdown else {'igemm_nt'}
    ops.KernelTimer.records = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3      # host enqueue time per step (launch-bound if ~ ms_per_step)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.KernelTimer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        out = {
            'metric': 'training images/sec/node', 'value': round(value, 1), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
            'data': 'synthetic',
            'config': {'workload': f'{args.model} 3x{size}x{size} synthetic training step '
                                   f'(fwd+loss+bwd+all-reduce+optimizer), per-GPU batch {args.batch}',
                       'model': args.model, 'global_batch': args.batch * world, 'per_gpu_batch': args.batch,
                       'parallelism': f'dp{world}', 'final_loss': round(final_loss, 4),
                       'loss_scale': scaler.get_scale(), 'host_enqueue_ms_per_step': round(host_ms, 3)},
            'model_mfma_frac': round(TRAIN_GFLOP_PER_IMG.get(args.model, 0) * value / world / 1e3 / PEAK_BF16_TFLOPS, 4),
        }
        summ = ops.KernelTimer.summary() if not args.no_kernel_timer else {}
        if 'igemm_nt' in summ:
            k = summ['igemm_nt']
            achieved = k['flops'] / (k['ms'] * 1e-3) / 1e12
            out['roofline'] = {'kernel': 'igemm_nt_kernel (implicit-GEMM conv / linear, fwd + dgrad)', 'bound': 'mfma',
                               'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(achieved / PEAK_BF16_TFLOPS, 4),
                               'traffic': pmc_traffic('igemm_nt') if args.model == 'resnet50' else None,
                               'launches': k['calls'], 'avg_launch_us': round(k['ms'] * 1e3 / k['calls'], 2)}
            out['kernel_breakdown_ms_per_step'] = {t: round(v['ms'] / args.steps, 3) for t, v in summ.items()}
            for t, v in summ.items():
                if v['bytes'] > 0:
                    out.setdefault('hbm_kernels', {})[t] = {
                        'GB/s': round(v['bytes'] / (v['ms'] * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(v['bytes'] / (v['ms'] * 1e-3) / 8e12, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.model)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
