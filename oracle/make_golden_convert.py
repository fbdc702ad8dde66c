"""Golden key mapping of the reference's torchvision-ResNet converter, produced by the reference's own tables:
SimpleAICV/classification/weight_convert/convert_resnet_weight_from_pytorch_offical_weight.py:15-74 (`convert_common_dict`,
`convert_other_dict`) applied with its own rule (:110-119: exact match in the common table first, else the first table entry
that is a substring of the key).  Run in the build container:
    python oracle/make_golden_convert.py   ->  tests/golden/convert_resnet_keys.json
The torchvision key list is generated from torchvision's naming rule (torchvision itself is not installed)."""
import ast
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/SimpleAICV/classification/weight_convert/convert_resnet_weight_from_pytorch_offical_weight.py'
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'convert_resnet_keys.json')

BN = ['weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked']


def torchvision_keys(blocks, bottleneck):
    keys = ['conv1.weight'] + [f'bn1.{s}' for s in BN]
    for li, nb in enumerate(blocks, 1):
        for b in range(nb):
            p = f'layer{li}.{b}.'
            for c in range(1, 4 if bottleneck else 3):
                keys.append(p + f'conv{c}.weight')
                keys += [p + f'bn{c}.{s}' for s in BN]
            if b == 0 and (bottleneck or li > 1):
                keys.append(p + 'downsample.0.weight')
                keys += [p + f'downsample.1.{s}' for s in BN]
    return keys + ['fc.weight', 'fc.bias']


def reference_tables():
    """The two module-level dict literals of the reference script, read without executing it (its __main__ part loads files)."""
    tree = ast.parse(open(REF).read())
    tables = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.value, ast.Dict) and node.targets[0].id in ('convert_common_dict', 'convert_other_dict'):
            tables[node.targets[0].id] = ast.literal_eval(node.value)
    return tables['convert_common_dict'], tables['convert_other_dict']


def reference_rule(key, common, other):
    if key in common:
        return common[key]
    for sub in other:
        if sub in key:
            return key.replace(sub, other[sub])
    return None


def reference_van_filter_list():
    """the module-level set literal `filter_list` of the reference's VAN converter (:14-37)"""
    path = os.path.join(os.path.dirname(REF), 'convert_van_weight_from_pytorch_offical_weight.py')
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.Assign) and node.targets[0].id == 'filter_list':
            return sorted(ast.literal_eval(node.value))
    raise RuntimeError('filter_list not found')


def main():
    json.dump({'filter_list': reference_van_filter_list()}, open(os.path.join(os.path.dirname(OUT), 'convert_van_filter.json'), 'w'), indent=0)
    common, other = reference_tables()
    out = {}
    for name, blocks, bott in (('resnet18', [2, 2, 2, 2], False), ('resnet50', [3, 4, 6, 3], True)):
        out[name] = [[k, reference_rule(k, common, other)] for k in torchvision_keys(blocks, bott)]
    json.dump(out, open(OUT, 'w'), indent=0)
    print({k: len(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
