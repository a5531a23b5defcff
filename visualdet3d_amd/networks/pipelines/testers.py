"""Test functions with the reference's signatures (``networks/pipelines/testers.py:15-42``): what ``cfg.trainer.test_func``
names and ``pipelines/evaluators.py:107`` calls with one collated frame."""
import torch

from ..utils.registry import PIPELINE_DICT


@PIPELINE_DICT.register_module
@torch.no_grad()
def test_mono_detection(data, module, writer=None, loss_logger=None, global_step=None, cfg=None):
    image, P2 = data[0], data[1]
    scores, bbox, obj_index = module([image.cuda().float().contiguous(), torch.as_tensor(P2).cuda().float()])
    return scores, bbox, [cfg.obj_types[int(i)] for i in obj_index.reshape(-1).tolist()]


@PIPELINE_DICT.register_module
@torch.no_grad()
def test_stereo_detection(data, module, writer=None, loss_logger=None, global_step=None, cfg=None):
    left, right, P2, P3 = data[0], data[1], data[2], data[3]
    scores, bbox, obj_index = module([left.cuda().float().contiguous(), right.cuda().float().contiguous(),
                                      torch.as_tensor(P2).cuda().float(), torch.as_tensor(P3).cuda().float()])
    return scores, bbox, [cfg.obj_types[int(i)] for i in obj_index.reshape(-1).tolist()]
