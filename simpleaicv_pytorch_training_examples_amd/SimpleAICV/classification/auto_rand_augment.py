"""AutoAugment / RandAugment on PIL images -- the loader-side policy transforms of the ViT / VAN / ConvFormer fine-tuning configs
(reference SimpleAICV/classification/auto_rand_augment.py: pixel ops :50-170, magnitude -> argument maps :173-256, AugmentOp :314-355,
the four AutoAugment policies :358-493, AutoAugment :538-565, RandAugment :646-691; itself the timm restatement of the published
AutoAugment (Cubuk et al. 2019) and RandAugment (Cubuk et al. 2020) policies).

Host work on one uint8 image per loader worker: PIL's C routines are the kernels here, nothing for the GPU to do before the batch
exists (the batch-level pieces -- uint8 collate, normalise, mixup / cutmix, RandomErasing -- are the device-side ones, csrc/input.hip).
What has to be identical to the reference is the DRAW ORDER on Python's `random` (and numpy's generator for RandAugment's op
choice), since the reference seeds both per worker:
    per op:   random.random()          apply or skip           (only when prob < 1)
              random.gauss / uniform   magnitude noise         (only when magnitude_std > 0)
              random.random()          sign of the argument    (signed ops only)
              random.choice            bilinear or bicubic     (geometric ops only, when no fixed interpolation is given)
tests/test_r04_host.py pins every op table entry, the four AutoAugment policies and RandAugment (uniform and weighted choice) to
images the reference produced under the same seeds (oracle/make_golden_r04.py auto_rand_augment)."""
import random

import numpy as np
from PIL import Image, ImageEnhance, ImageOps

_LEVEL_DENOM = 10.          # a magnitude M means the fraction M / 10 of an op's range
_FILL = (128, 128, 128)
_HPARAMS_DEFAULT = dict(translate_const=250, img_mean=_FILL)
_RANDOM_INTERPOLATION = (Image.Resampling.BILINEAR, Image.Resampling.BICUBIC)


# ------------------------------------------------------------------------------------------------------------ pixel operations
def _affine(coeffs):
    """geometric op: PIL affine transform with the coefficients `coeffs(img, v)`; picks the interpolation (one draw when a
    tuple of candidates is configured) before transforming, as the reference's _check_args_tf does"""
    def op(img, v, **kw):
        kw['resample'] = _pick_interpolation(kw)
        return img.transform(img.size, Image.AFFINE, coeffs(img, v), **kw)
    return op


def _pick_interpolation(kw):
    how = kw.pop('resample', Image.Resampling.BILINEAR)
    return random.choice(how) if isinstance(how, (list, tuple)) else how


shear_x = _affine(lambda img, f: (1, f, 0, 0, 1, 0))
shear_y = _affine(lambda img, f: (1, 0, 0, f, 1, 0))
translate_x_rel = _affine(lambda img, pct: (1, 0, pct * img.size[0], 0, 1, 0))
translate_y_rel = _affine(lambda img, pct: (1, 0, 0, 0, 1, pct * img.size[1]))
translate_x_abs = _affine(lambda img, px: (1, 0, px, 0, 1, 0))
translate_y_abs = _affine(lambda img, px: (1, 0, 0, 0, 1, px))


def rotate(img, degrees, **kw):
    kw['resample'] = _pick_interpolation(kw)
    return img.rotate(degrees, **kw)            # Pillow >= 5.2 (this image: 12.x) takes fillcolor / resample directly


def auto_contrast(img, **_):
    return ImageOps.autocontrast(img)


def invert(img, **_):
    return ImageOps.invert(img)


def equalize(img, **_):
    return ImageOps.equalize(img)


def solarize(img, thresh, **_):
    return ImageOps.solarize(img, thresh)


def solarize_add(img, add, thresh=128, **_):
    if img.mode not in ('L', 'RGB'):
        return img
    lut = [min(255, v + add) if v < thresh else v for v in range(256)]
    return img.point(lut * (3 if img.mode == 'RGB' else 1))


def posterize(img, bits_to_keep, **_):
    return img if bits_to_keep >= 8 else ImageOps.posterize(img, bits_to_keep)


def _enhancer(kind):
    def op(img, factor, **_):
        return kind(img).enhance(factor)
    return op


contrast, color = _enhancer(ImageEnhance.Contrast), _enhancer(ImageEnhance.Color)
brightness, sharpness = _enhancer(ImageEnhance.Brightness), _enhancer(ImageEnhance.Sharpness)


# ----------------------------------------------------------------------------------------------- magnitude -> op argument
def _randomly_negate(v):
    return -v if random.random() > 0.5 else v


def _signed(span):
    """symmetric range [-span, span], the sign drawn"""
    return lambda level, hp: (_randomly_negate((level / _LEVEL_DENOM) * (span(hp) if callable(span) else span)),)


def _enhance(level, _hp):                              # [0.1, 1.9]
    return ((level / _LEVEL_DENOM) * 1.8 + 0.1,)


def _enhance_increasing(level, _hp):                   # 1.0 is "no change"; strength grows either way, floor 0.1
    return (max(0.1, 1.0 + _randomly_negate((level / _LEVEL_DENOM) * .9)),)


def _bits(level):
    return int((level / _LEVEL_DENOM) * 4)


def _thresh(level):
    return int((level / _LEVEL_DENOM) * 256)


# name -> (pixel op, magnitude map or None)
_OPS = {
    'AutoContrast': (auto_contrast, None),
    'Equalize': (equalize, None),
    'Invert': (invert, None),
    'Rotate': (rotate, _signed(30.)),
    'Posterize': (posterize, lambda level, hp: (_bits(level),)),                          # keep 0..4 bits (TPU EfficientNet)
    'PosterizeIncreasing': (posterize, lambda level, hp: (4 - _bits(level),)),            # keep 4..0 bits
    'PosterizeOriginal': (posterize, lambda level, hp: (_bits(level) + 4,)),              # keep 4..8 bits (the paper)
    'Solarize': (solarize, lambda level, hp: (_thresh(level),)),
    'SolarizeIncreasing': (solarize, lambda level, hp: (256 - _thresh(level),)),
    'SolarizeAdd': (solarize_add, lambda level, hp: (int((level / _LEVEL_DENOM) * 110),)),
    'Color': (color, _enhance), 'ColorIncreasing': (color, _enhance_increasing),
    'Contrast': (contrast, _enhance), 'ContrastIncreasing': (contrast, _enhance_increasing),
    'Brightness': (brightness, _enhance), 'BrightnessIncreasing': (brightness, _enhance_increasing),
    'Sharpness': (sharpness, _enhance), 'SharpnessIncreasing': (sharpness, _enhance_increasing),
    'ShearX': (shear_x, _signed(0.3)), 'ShearY': (shear_y, _signed(0.3)),
    'TranslateX': (translate_x_abs, _signed(lambda hp: float(hp['translate_const']))),
    'TranslateY': (translate_y_abs, _signed(lambda hp: float(hp['translate_const']))),
    'TranslateXRel': (translate_x_rel, _signed(lambda hp: hp.get('translate_pct', 0.45))),
    'TranslateYRel': (translate_y_rel, _signed(lambda hp: hp.get('translate_pct', 0.45))),
}
NAME_TO_OP = {k: v[0] for k, v in _OPS.items()}
LEVEL_TO_ARG = {k: v[1] for k, v in _OPS.items()}


class AugmentOp:
    """one policy entry: apply `name` with probability `prob` at `magnitude` (optionally noised, clipped to [0, magnitude_max or 10])"""

    def __init__(self, name, prob=0.5, magnitude=10, hparams=None):
        hparams = hparams or _HPARAMS_DEFAULT
        self.name, self.prob, self.magnitude = name, prob, magnitude
        self.aug_fn, self.level_fn = _OPS[name]
        self.hparams = hparams.copy()
        self.kwargs = dict(fillcolor=hparams.get('img_mean', _FILL), resample=hparams.get('interpolation', _RANDOM_INTERPOLATION))
        self.magnitude_std = self.hparams.get('magnitude_std', 0)       # inf: uniform in [0, magnitude]
        self.magnitude_max = self.hparams.get('magnitude_max', None)

    def __call__(self, img):
        if self.prob < 1.0 and random.random() > self.prob:
            return img
        m = self.magnitude
        if self.magnitude_std > 0:
            m = random.uniform(0, m) if self.magnitude_std == float('inf') else random.gauss(m, self.magnitude_std)
        m = max(0., min(m, self.magnitude_max or _LEVEL_DENOM))
        args = self.level_fn(m, self.hparams) if self.level_fn is not None else ()
        return self.aug_fn(img, *args, **self.kwargs)


# ------------------------------------------------------------------------------------------------------- AutoAugment policies
# 25 sub-policies of two (op, probability, magnitude) entries each, written "Op p m, Op p m; ...".  'v0' is the TPU EfficientNet
# ImageNet policy, 'original' the paper's; the '...r' forms swap their Posterize variant for PosterizeIncreasing.
_POLICY_TEXT = {
    'v0': 'Equalize .8 1, ShearY .8 4; Color .4 9, Equalize .6 3; Color .4 1, Rotate .6 8; Solarize .8 3, Equalize .4 7; '
          'Solarize .4 2, Solarize .6 2; Color .2 0, Equalize .8 8; Equalize .4 8, SolarizeAdd .8 3; ShearX .2 9, Rotate .6 8; '
          'Color .6 1, Equalize 1 2; Invert .4 9, Rotate .6 0; Equalize 1 9, ShearY .6 3; Color .4 7, Equalize .6 0; '
          'Posterize .4 6, AutoContrast .4 7; Solarize .6 8, Color .6 9; Solarize .2 4, Rotate .8 9; Rotate 1 7, TranslateYRel .8 9; '
          'ShearX 0 0, Solarize .8 4; ShearY .8 0, Color .6 4; Color 1 0, Rotate .6 2; Equalize .8 4, Equalize 0 8; '
          'Equalize 1 4, AutoContrast .6 2; ShearY .4 7, SolarizeAdd .6 7; Posterize .8 2, Solarize .6 10; '
          'Solarize .6 8, Equalize .6 1; Color .8 6, Rotate .4 5',
    'original': 'PosterizeOriginal .4 8, Rotate .6 9; Solarize .6 5, AutoContrast .6 5; Equalize .8 8, Equalize .6 3; '
                'PosterizeOriginal .6 7, PosterizeOriginal .6 6; Equalize .4 7, Solarize .2 4; Equalize .4 4, Rotate .8 8; '
                'Solarize .6 3, Equalize .6 7; PosterizeOriginal .8 5, Equalize 1 2; Rotate .2 3, Solarize .6 8; '
                'Equalize .6 8, PosterizeOriginal .4 6; Rotate .8 8, Color .4 0; Rotate .4 9, Equalize .6 2; '
                'Equalize 0 7, Equalize .8 8; Invert .6 4, Equalize 1 8; Color .6 4, Contrast 1 8; Rotate .8 8, Color 1 2; '
                'Color .8 8, Solarize .8 7; Sharpness .4 7, Invert .6 8; ShearX .6 5, Equalize 1 9; Color .4 0, Equalize .6 3; '
                'Equalize .4 7, Solarize .2 4; Solarize .6 5, AutoContrast .6 5; Invert .6 4, Equalize 1 8; '
                'Color .6 4, Contrast 1 8; Equalize .8 8, Equalize .6 3',
}
_POLICY_TEXT['v0r'] = _POLICY_TEXT['v0'].replace('Posterize ', 'PosterizeIncreasing ')
_POLICY_TEXT['originalr'] = _POLICY_TEXT['original'].replace('PosterizeOriginal ', 'PosterizeIncreasing ')


def auto_augment_policy(name='v0', hparams=None):
    hparams = hparams or _HPARAMS_DEFAULT
    assert name in _POLICY_TEXT, 'Unknown AA policy (%s)' % name
    policy = []
    for pair in _POLICY_TEXT[name].split(';'):
        entries = [e.split() for e in pair.split(',')]
        policy.append([AugmentOp(op, float(p), int(m), hparams=hparams) for op, p, m in entries])
    return policy


def _hparams(resize, mean, magnitude_std=None, magnitude_max=None):
    hp = dict(translate_const=int(resize * 0.45), img_mean=tuple(min(255, round(255 * x)) for x in mean))
    if magnitude_std:
        hp['magnitude_std'] = float(magnitude_std)
    if magnitude_max:
        hp['magnitude_max'] = int(magnitude_max)
    return hp


class AutoAugment:
    """sample {'image': PIL image, 'label': ...}: one of the policy's 25 sub-policies (random.choice), its two ops in order"""

    def __init__(self, policy_name, resize=224, mean=[0.485, 0.456, 0.406], magnitude_std=None):
        assert policy_name in ['original', 'originalr', 'v0', 'v0r']
        self.policy = auto_augment_policy(policy_name, hparams=_hparams(resize, mean, magnitude_std))

    def __call__(self, sample):
        image = sample['image']
        for op in random.choice(self.policy):
            image = op(image)
        sample['image'] = image
        return sample


_RAND_TRANSFORMS = ['AutoContrast', 'Equalize', 'Invert', 'Rotate', 'Posterize', 'Solarize', 'SolarizeAdd', 'Color', 'Contrast',
                    'Brightness', 'Sharpness', 'ShearX', 'ShearY', 'TranslateXRel', 'TranslateYRel']
_INCREASING = {'Posterize', 'Solarize', 'Color', 'Contrast', 'Brightness', 'Sharpness'}
_RAND_INCREASING_TRANSFORMS = [n + 'Increasing' if n in _INCREASING else n for n in _RAND_TRANSFORMS]
# experimental choice weights (weight_idx 0), in the order of _RAND_TRANSFORMS
_RAND_CHOICE_WEIGHTS_0 = dict(zip(_RAND_TRANSFORMS, (.025, .005, 0, .3, 0, .005, .005, .025, .005, .005, .025, .2, .2, .1, .1)))


def _select_rand_weights(weight_idx=0, transforms=None):
    assert weight_idx == 0          # only one set of weights exists
    probs = np.array([_RAND_CHOICE_WEIGHTS_0[k] for k in (transforms or _RAND_TRANSFORMS)])
    return probs / np.sum(probs)


def rand_augment_ops(magnitude=10, hparams=None, transforms=None):
    return [AugmentOp(name, prob=0.5, magnitude=magnitude, hparams=hparams or _HPARAMS_DEFAULT) for name in (transforms or _RAND_TRANSFORMS)]


class RandAugment:
    """sample {'image': PIL image, 'label': ...}: `num_layers` ops drawn by numpy (with replacement when unweighted, without when
    weight_idx is given), each applied with probability 0.5 at `magnitude` (+ gaussian noise of magnitude_std)"""

    def __init__(self, magnitude=9, num_layers=2, resize=224, mean=[0.485, 0.456, 0.406], integer=True, weight_idx=None,
                 magnitude_std=0.5, magnitude_max=None):
        self.ops = rand_augment_ops(magnitude=magnitude, hparams=_hparams(resize, mean, magnitude_std, magnitude_max),
                                    transforms=_RAND_INCREASING_TRANSFORMS if integer else _RAND_TRANSFORMS)
        self.num_layers = num_layers
        self.choice_weights = None if weight_idx is None else _select_rand_weights(weight_idx)

    def __call__(self, sample):
        image = sample['image']
        for op in np.random.choice(self.ops, self.num_layers, replace=self.choice_weights is None, p=self.choice_weights):
            image = op(image)
        sample['image'] = image
        return sample
