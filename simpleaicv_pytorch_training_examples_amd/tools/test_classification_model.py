"""torchrun entry point for classification evaluation on MI355X -- same CLI (`--work-dir`), same `test_config.py` contract and the
same two log lines as the reference tools/test_classification_model.py (:31-103):

    torchrun --nproc_per_node=N --master_addr 127.0.0.1 --master_port P \\
        -m simpleaicv_pytorch_training_examples_amd.tools.test_classification_model --work-dir ./

    model: <network>, flops: ..., macs: ..., params: ...
    acc1: ..%, acc5: ..%, test_loss: .., per_image_load_time: ..ms, per_image_inference_time: ..ms

One process per GPU, DistributedSampler over the test set, the engine's DDP wrapper around the model (evaluation has no gradient
traffic; the wrapper is there for the `module.` state-dict surface and the reduction group test_classification uses)."""
import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader

from .. import engine
from .scripts import test_classification
from .utils import compute_macs_and_params, get_logger, set_seed


def parse_args():
    parser = argparse.ArgumentParser(description='PyTorch Classification Testing (MI355X engine)')
    parser.add_argument('--work-dir', type=str, help='path for get testing config')
    return parser.parse_args()


def main():
    assert torch.cuda.is_available(), 'need gpu to train network!'
    args = parse_args()
    sys.path.append(args.work_dir)
    from test_config import config
    log_dir = os.path.join(args.work_dir, 'log')
    config.gpus_type = torch.cuda.get_device_name()
    config.gpus_num = int(os.environ.get('WORLD_SIZE', torch.cuda.device_count()))
    set_seed(config.seed)
    local_rank = int(os.environ['LOCAL_RANK'])
    config.local_rank = local_rank
    torch.cuda.set_device(local_rank)
    torch.distributed.init_process_group(backend='nccl', init_method='env://', device_id=torch.device('cuda', local_rank))
    config.group = torch.distributed.new_group(list(range(config.gpus_num)))
    os.makedirs(log_dir, exist_ok=True)
    torch.distributed.barrier(device_ids=[local_rank])
    logger = get_logger('test', log_dir)
    info = (lambda m: logger.info(m)) if local_rank == 0 else (lambda m: None)

    assert config.batch_size % config.gpus_num == 0, 'config.batch_size is not divisible by config.gpus_num!'
    assert config.num_workers % config.gpus_num == 0, 'config.num_workers is not divisible by config.gpus_num!'
    batch_size = int(config.batch_size // config.gpus_num)
    num_workers = int(config.num_workers // config.gpus_num)
    test_sampler = torch.utils.data.distributed.DistributedSampler(config.test_dataset, shuffle=False)
    test_loader = DataLoader(config.test_dataset, batch_size=batch_size, shuffle=False, pin_memory=True, num_workers=num_workers,
                             collate_fn=config.test_collater, sampler=test_sampler)
    for key, value in config.__dict__.items():
        if not key.startswith('__') and key not in ['model']:
            info(f'{key}: {value}')

    model, test_criterion = config.model, config.test_criterion
    flops, macs, params = compute_macs_and_params(config, model)
    info(f'model: {config.network}, flops: {flops}, macs: {macs}, params: {params}')
    model = model.cuda()
    test_criterion = test_criterion.cuda()
    model = engine.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank, process_group=config.group)
    acc1, acc5, test_loss, per_image_load_time, per_image_inference_time = test_classification(test_loader, model, test_criterion,
                                                                                              config)
    info(f'acc1: {acc1:.3f}%, acc5: {acc5:.3f}%, test_loss: {test_loss:.4f}, per_image_load_time: {per_image_load_time:.3f}ms, '
         f'per_image_inference_time: {per_image_inference_time:.3f}ms')
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
