"""torchrun entry point for SAM training on MI355X -- same CLI (`--work-dir`), same `train_config.py`
contract, log lines and checkpoint schema as the reference tools/train_interactive_segmentation_model.py
(:33-258), launched the same way:

    torchrun --nproc_per_node=N --master_addr 127.0.0.1 --master_port P \\
        -m simpleaicv_pytorch_training_examples_amd.tools.train_interactive_segmentation_model --work-dir ./

One process per GPU; backend "nccl" (= RCCL over xGMI on ROCm).  Checkpoints: checkpoints/latest.pth =
{epoch, time, best_loss, train_loss, lr, model_state_dict (`module.`-prefixed), optimizer_state_dict,
scheduler_state_dict}; epoch_N.pth every save_interval; best weights -> `{network}-loss{best:.3f}.pth`.
"""
import argparse
import functools
import os
import sys
import time

# (a captured step needs ROCm's graph packet capture OFF, and the HIP runtime reads that switch before its first call: running this
# module with `-m` imports the package -- whose __init__ sets it -- before torch is imported below)
import torch
from torch.utils.data import DataLoader

from .interactive_segmentation_scripts import train_sam_segmentation
from .utils import Scheduler, build_optimizer, build_training_mode, get_logger, set_seed, worker_seed_init_fn


def parse_args():
    parser = argparse.ArgumentParser(description='PyTorch Interactive Segmentation Training (MI355X engine)')
    parser.add_argument('--work-dir', type=str, help='path for get training config and saving log/models')
    return parser.parse_args()


def main():
    assert torch.cuda.is_available(), 'need gpu to train network!'
    args = parse_args()
    sys.path.append(args.work_dir)
    from train_config import config
    log_dir = os.path.join(args.work_dir, 'log')
    checkpoint_dir = os.path.join(args.work_dir, 'checkpoints')
    resume_model = os.path.join(checkpoint_dir, 'latest.pth')
    config.gpus_type = torch.cuda.get_device_name()
    config.gpus_num = int(os.environ.get('WORLD_SIZE', torch.cuda.device_count()))
    set_seed(config.seed)
    local_rank = int(os.environ['LOCAL_RANK'])
    config.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    torch.distributed.init_process_group(backend='nccl', init_method='env://',
                                         device_id=torch.device('cuda', local_rank))
    config.group = torch.distributed.new_group(list(range(config.gpus_num)))
    os.makedirs(checkpoint_dir, exist_ok=True)
    os.makedirs(log_dir, exist_ok=True)
    torch.distributed.barrier(device_ids=[local_rank])
    logger = get_logger('train', log_dir)
    info = (lambda m: logger.info(m)) if local_rank == 0 else (lambda m: None)

    assert config.batch_size % config.gpus_num == 0, 'config.batch_size is not divisible by config.gpus_num!'
    assert config.num_workers % config.gpus_num == 0, 'config.num_workers is not divisible by config.gpus_num!'
    batch_size = int(config.batch_size // config.gpus_num)
    num_workers = int(config.num_workers // config.gpus_num)
    init_fn = functools.partial(worker_seed_init_fn, num_workers=num_workers, local_rank=local_rank, seed=config.seed)
    train_sampler = torch.utils.data.distributed.DistributedSampler(config.train_dataset, shuffle=True)
    train_loader = DataLoader(config.train_dataset, batch_size=batch_size, shuffle=False, pin_memory=True,
                              drop_last=True, num_workers=num_workers, collate_fn=config.train_collater,
                              sampler=train_sampler, worker_init_fn=init_fn)

    for key, value in config.__dict__.items():
        if not key.startswith('__') and key not in ['model']:
            info(f'{key}: {value}')

    model = config.model.cuda()
    train_criterion = config.train_criterion.cuda()
    info('--------------------parameters--------------------')
    for name, param in model.named_parameters():
        info(f'name: {name}, grad: {param.requires_grad}')
    info('--------------------buffers--------------------')
    for name, buffer in model.named_buffers():
        info(f'name: {name}, grad: {buffer.requires_grad}')

    optimizer, model_layer_weight_decay_list = build_optimizer(config, model)
    info('-------------layers weight decay---------------')
    for per_layer_list in model_layer_weight_decay_list:
        lr_scale = per_layer_list.get('lr_scale', 'not setting!')
        for name in per_layer_list['name']:
            info(f"name: {name}, lr: {per_layer_list['lr']}, weight_decay: {per_layer_list['weight_decay']}, "
                 f'lr_scale: {lr_scale}')

    scheduler = Scheduler(config, optimizer)
    model, _, config.scaler = build_training_mode(config, model)

    start_epoch, train_time = 1, 0
    best_loss, train_loss = 1e9, 0
    if os.path.exists(resume_model):
        checkpoint = torch.load(resume_model, map_location=torch.device('cpu'), weights_only=True)
        model.load_state_dict(checkpoint['model_state_dict'])
        optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        scheduler.load_state_dict(checkpoint['scheduler_state_dict'])
        saved_epoch = checkpoint['epoch']
        start_epoch += saved_epoch
        used_time = checkpoint['time']
        train_time += used_time
        best_loss, train_loss, lr = checkpoint['best_loss'], checkpoint['train_loss'], checkpoint['lr']
        info(f'resuming model from {resume_model}. resume_epoch: {saved_epoch:0>3d}, used_time: {used_time:.3f} hours, '
             f'best_loss: {best_loss:.4f}, lr: {lr:.6f}')
        from .. import ops
        ops.bump_weights_epoch()

    info(f'using torch version:{torch.__version__}')
    config.compile_support = False      # torch.compile (Inductor -> Triton) is not part of the MI355X-native path
    config.use_compile = False

    for epoch in range(start_epoch, config.epochs + 1):
        per_epoch_start_time = time.time()
        info(f'epoch {epoch:0>3d} lr: {scheduler.current_lr:.6f}')
        train_sampler.set_epoch(epoch)
        train_loss = train_sam_segmentation(train_loader, model, train_criterion, optimizer, scheduler, epoch, logger,
                                            config)
        info(f'train: epoch {epoch:0>3d}, train_loss: {train_loss:.4f}')
        train_time += (time.time() - per_epoch_start_time) / 3600
        if local_rank == 0:
            if epoch % config.save_interval == 0:
                torch.save(model.module.state_dict(), os.path.join(checkpoint_dir, f'epoch_{epoch}.pth'))
            if train_loss < best_loss:
                best_loss = train_loss
                torch.save(model.module.state_dict(), os.path.join(checkpoint_dir, 'best.pth'))
            torch.save({'epoch': epoch, 'time': train_time, 'best_loss': best_loss, 'train_loss': train_loss,
                        'lr': scheduler.current_lr, 'model_state_dict': model.state_dict(),
                        'optimizer_state_dict': optimizer.state_dict(), 'scheduler_state_dict': scheduler.state_dict()},
                       os.path.join(checkpoint_dir, 'latest.pth'))
        info(f'until epoch: {epoch:0>3d}, best_loss: {best_loss:.4f}')

    if local_rank == 0 and os.path.exists(os.path.join(checkpoint_dir, 'best.pth')):
        os.rename(os.path.join(checkpoint_dir, 'best.pth'),
                  os.path.join(checkpoint_dir, f'{config.network}-loss{best_loss:.3f}.pth'))
    info(f'train done. model: {config.network}, train time: {train_time:.3f} hours, best_loss: {best_loss:.4f}')
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
