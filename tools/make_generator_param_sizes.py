"""Writes tools/generator_param_sizes.json: name and shape of every parameter of the live reference Generator (cub
configuration: latent 512, attention_values 10, use_sdf), in registration order.  tools/train_bench.py pads its stand-in
plane producer with tensors of exactly these sizes, so that the gradient buckets of the training benchmark have the
reference's layout (129 tensors from 1 to 2 359 296 elements, models/stylegan.py:438-490) instead of one ballast tensor.
Needs /root/reference (run here, not on the GPU box); only names and shapes are recorded."""
import json
import os
import sys

sys.path.insert(0, '/root/reference')
from models import generator as g  # noqa: E402

m = g.Generator(512, 2.0, attention_values=10, use_sdf=True)
out = [{'name': n, 'shape': list(p.shape)} for n, p in m.named_parameters()]
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'generator_param_sizes.json')
json.dump({'source': 'models/generator.py Generator(512, 2.0, attention_values=10, use_sdf=True)', 'parameters': out},
          open(path, 'w'), indent=0)
print(len(out), 'tensors,', sum(int(__import__('math').prod(o['shape'])) for o in out), 'parameters ->', path)
