"""Synthetic stand-in for the reference's classification datasets (SimpleAICV/classification/datasets/
ilsvrc2012dataset.py, cifar100dataset.py) in the benchmark configs: there is no dataset, OpenCV or torchvision in
the build / bench images.  A sample has the contract the reference's datasets hand to the collater AFTER their
transform pipeline: {'image': float32 HWC array with post-normalisation statistics (N(0, 1)), 'label': int}.
Samples are generated per index from a counter-based RNG, so any worker / rank sees the same sample i."""
import numpy as np
from torch.utils.data import Dataset


class SyntheticClassificationDataset(Dataset):

    def __init__(self, num_samples, image_size, num_classes, seed=0, transform=None):
        self.num_samples, self.image_size, self.num_classes, self.seed = num_samples, image_size, num_classes, seed
        self.transform = transform

    def __len__(self):
        return self.num_samples

    def __getitem__(self, idx):
        rng = np.random.default_rng((self.seed, idx))
        image = rng.standard_normal((self.image_size, self.image_size, 3), dtype=np.float32)
        sample = {'image': image, 'label': int(rng.integers(0, self.num_classes))}
        if self.transform is not None:
            sample = self.transform(sample)
        return sample
