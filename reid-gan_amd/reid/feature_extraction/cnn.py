"""extract_cnn_feature — FD-GAN-master/reid/feature_extraction/cnn.py:9-27 on the HIP modules: eval-mode forward of a batch
(BASELINE config 1: `create('resnet50', cut_at_pooling=True)` on 64 crops -> [64, 2048]); the optional `modules` hooks return
the named sub-modules' outputs like the reference."""
from __future__ import absolute_import

from collections import OrderedDict

import torch


def extract_cnn_feature(model, inputs, modules=None):
    model.eval()
    if not torch.is_tensor(inputs):
        inputs = torch.as_tensor(inputs)
    inputs = inputs.to(torch.device("cuda", torch.cuda.current_device()), non_blocking=True)
    with torch.no_grad():
        if modules is None:
            return model(inputs).data.cpu()
        outputs = OrderedDict((id(m), None) for m in modules)
        handles = [m.register_forward_hook(lambda mod, i, o: outputs.__setitem__(id(mod), o.data.cpu())) for m in modules]
        try:
            model(inputs)
        finally:
            for h in handles:
                h.remove()
    return list(outputs.values())
