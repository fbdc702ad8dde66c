"""Synthetic CIFAR-100 pickles for BASELINE.json configs[0] (there is no network for the real ones): 50 000 x 3072 uint8 with
class-dependent means so that one epoch has something to learn, 100 fine labels, `meta`; deterministic (RandomState(0)), so the
reference's host run (oracle/run_reference_cifar_epoch.py) and the engine's entry script (SAICV_CIFAR_PICKLES=<dir> in
00.classification_training/cifar100/resnet18cifar/train_config.py) read byte-identical files.
    python scripts/cifar_synthetic_pickles.py <dir>"""
import os
import pickle
import sys

import numpy as np

N_TRAIN, N_TEST, CLASSES = 50000, 2048, 100


def write_pickles(d, with_test=False):
    rng = np.random.RandomState(0)
    labels = rng.randint(0, CLASSES, size=N_TRAIN)
    base = rng.randint(40, 216, size=(CLASSES, 3072)).astype(np.int16)
    data = np.clip(base[labels] + rng.randint(-48, 49, size=(N_TRAIN, 3072)), 0, 255).astype(np.uint8)
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, 'train'), 'wb') as f:
        pickle.dump({'data': data, 'fine_labels': labels.tolist()}, f)
    with open(os.path.join(d, 'meta'), 'wb') as f:
        pickle.dump({'fine_label_names': [f'class_{i}' for i in range(CLASSES)]}, f)
    if with_test:       # drawn AFTER the training set, so the training bytes do not depend on this flag
        tl = rng.randint(0, CLASSES, size=N_TEST)
        td = np.clip(base[tl] + rng.randint(-48, 49, size=(N_TEST, 3072)), 0, 255).astype(np.uint8)
        with open(os.path.join(d, 'test'), 'wb') as f:
            pickle.dump({'data': td, 'fine_labels': tl.tolist()}, f)


class Normalize:
    """[H, W, 3] float32 0..255 -> (x / 255 - mean) / std, the constants of the reference config
    (00.classification_training/cifar100/resnet18cifar/train_config.py:55-60)"""
    mean = np.array([0.5071, 0.4865, 0.4409], dtype=np.float32)
    std = np.array([0.2673, 0.2564, 0.2762], dtype=np.float32)

    def __call__(self, sample):
        sample['image'] = (sample['image'] / 255.0 - self.mean) / self.std
        return sample


if __name__ == '__main__':
    write_pickles(sys.argv[1], with_test=True)
    print('wrote', sorted(os.listdir(sys.argv[1])))
