"""Module-level FLAGS namespace.

Mirrors the ``tf.app.flags`` globals the reference reads inside library code
(/root/reference/atvsnet/example.py:25-48; read at homography_warping.py:149,
215,301,321,369,378, model.py:96,248).  Same attribute names and defaults, so
call sites stay drop-in.
"""


class _Flags(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.root_path = '../example/'
        self.example_index = 2
        self.view_num = 5
        self.pretrained_model_ckpt_path = '../model/model.ckpt'
        self.max_d = 128
        self.num_gpus = 1
        self.gpu_id = 0
        self.sample_scale = 0.25
        self.batch_size = 1
        self.inverse_depth = True
        self.synthetic_weights = False   # not a reference flag: seeded random weights instead of a checkpoint (example.cli)
        # eval_pointcloud.py (reference :31-57; its own defaults for view_num / max_d are set by its cli)
        self.data_root = '../data/'
        self.savepath = '../eval/pointcloud/'
        self.max_w = 896
        self.max_h = 480
        self.adaptive_scaling = True


FLAGS = _Flags()

# stand-in for tf.AUTO_REUSE (model.py:132 ...): variables are always shared by name here
AUTO_REUSE = 'AUTO_REUSE'
