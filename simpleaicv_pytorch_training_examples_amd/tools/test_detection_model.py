"""torchrun entry point for detection evaluation on MI355X -- same CLI (`--work-dir`), same `test_config.py` contract and log lines as
the reference tools/test_detection_model.py (:29-98):

    torchrun --nproc_per_node=N --master_addr 127.0.0.1 --master_port P \\
        -m simpleaicv_pytorch_training_examples_amd.tools.test_detection_model --work-dir ./

    model: <network>, flops: ..., macs: ..., params: ...
    eval type: COCO | VOC, then one `key: value` line per entry of the result dict

As in the reference the test loader is NOT sharded (every rank decodes the whole set; rank 0 logs).  COCO-style numbers come from
tools/cocoeval_numpy.py (pycocotools is not in the image), VOC-style ones from the per-class AP of tools/scripts.py."""
import argparse
import os
import sys

import torch
from torch.utils.data import DataLoader

from .. import engine
from .scripts import test_detection
from .utils import compute_macs_and_params, get_logger, set_seed


def parse_args():
    parser = argparse.ArgumentParser(description='PyTorch Detection Testing (MI355X engine)')
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
    test_loader = DataLoader(config.test_dataset, batch_size=batch_size, shuffle=False, pin_memory=True, num_workers=num_workers,
                             collate_fn=config.test_collater)
    for key, value in config.__dict__.items():
        if not key.startswith('__') and key not in ['model']:
            info(f'{key}: {value}')

    model, test_criterion, decoder = config.model, config.test_criterion, config.decoder
    flops, macs, params = compute_macs_and_params(config, model)
    info(f'model: {config.network}, flops: {flops}, macs: {macs}, params: {params}')
    model = model.cuda()
    test_criterion = test_criterion.cuda()
    model = engine.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank, process_group=config.group)
    result_dict = test_detection(test_loader, model, test_criterion, decoder, config)
    log_info = f'eval type: {config.eval_type}\n'
    for key, value in result_dict.items():
        log_info += f'{key}: {value}\n'
    info(log_info)
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
