"""CIFAR-100 in its published "python" pickle format (reference SimpleAICV/classification/datasets/cifar100dataset.py:12-64): the
files `<root_dir>/train` | `test` hold {'data': uint8 [N, 3072] (channel planes R, G, B of 32 x 32), 'fine_labels': [N]}, `meta`
holds 'fine_label_names'.  A sample is what the reference hands its transform chain: {'image': float32 [32, 32, 3] in 0..255,
'label': float32 scalar}; `class_name_to_label` / `label_to_class_name` as there.  Host-side reader, no device code."""
import os
import pickle

import numpy as np
from torch.utils.data import Dataset


class CIFAR100Dataset(Dataset):

    def __init__(self, root_dir, set_name='train', transform=None):
        assert set_name in ['train', 'test'], 'Wrong set name!'
        with open(os.path.join(root_dir, set_name), 'rb') as f:
            blob = pickle.load(f, encoding='latin1')
        with open(os.path.join(root_dir, 'meta'), 'rb') as f:
            names = pickle.load(f, encoding='latin1')['fine_label_names']
        planes = np.asarray(blob['data'])
        self.images = planes.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)      # [N, H, W, C] view of the channel planes
        self.labels = np.asarray(blob['fine_labels'])
        self.class_name_to_label = {name: i for i, name in enumerate(names)}
        self.label_to_class_name = {i: name for i, name in enumerate(names)}
        self.transform = transform
        print(f'Dataset Size:{self.images.shape[0]}')
        print(f'Dataset Class Num:{len(self.class_name_to_label)}')

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, idx):
        sample = {'image': np.array(self.images[idx]).astype(np.float32), 'label': np.array(self.labels[idx]).astype(np.float32)}
        if self.transform:
            sample = self.transform(sample)
        return sample
