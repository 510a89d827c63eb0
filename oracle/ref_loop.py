"""TEST INFRASTRUCTURE ONLY, BUILD CONTAINER ONLY -- runs the reference's OWN training loop on the CPU.

`run(...)` imports /root/reference's `depth_fine_tuning.DepthFineTuner`, `loaders.video_dataset`, `loss.joint_loss`,
`monodepth.*` UNMODIFIED (nothing is copied; the checkout is read-only and does not exist on the GPU box) and calls
`DepthFineTuner(range_dir, frames, params).fine_tune()` and `.save_depth()` on an on-disk clip.  What the container
lacks is stubbed through `sys.modules` (SURVEY.md section 8c lists the set):

  * `cv2` (mask PNGs are read with PIL; colour-mapped PNG previews become no-ops), `wget`, `torchvision.utils`,
    `torch.utils.tensorboard` (no-op writer);
  * the un-vendored network packages.  `monodepth.mannequin_challenge.models.pix2pix_model.Pix2PixModel` is given
    the interface `monodepth/mannequin_challenge_model.py:34-50,60,72` uses (`load_network`, `netG`,
    `switch_to_train/eval`); its `netG` is oracle/hourglass_ref.py (the restated hourglass, "parity unpinned" --
    see that file) over the state dict the caller provides as the "downloaded" checkpoint `checkpoints/mc.pth`.

Two environment adaptations, neither touching the reference's arithmetic: `Tensor.cuda()` is the identity (the
reference hard-codes `.cuda()` at mannequin_challenge_model.py:53; there is no GPU here), and for the fp64 run
`loaders.video_dataset._dtype` (the module constant the loaders and JointLoss read) is set to float64 before the
loss modules import it.  The step order of the shuffled DataLoader is read off the loop's own log lines
("Epoch = e, pairs = [[i, j], ...], loss = v", depth_fine_tuning.py:275) so that another loop can replay it.
"""
import argparse
import contextlib
import io
import os
import re
import sys
import types

import numpy as np
import torch

from . import hourglass_ref

REF = "/root/reference"
_REF_TOP = ("utils", "loss", "loaders", "monodepth", "optimizer", "depth_fine_tuning")


def available():
    return os.path.isfile(os.path.join(REF, "depth_fine_tuning.py"))


class _RefNet(torch.nn.Module):
    """hourglass_ref.forward as the `netG` module: parameters / buffers keyed like the upstream checkpoint."""

    def __init__(self, state_dict, dtype):
        super().__init__()
        self._keys = list(state_dict)
        self._state = {}
        for i, k in enumerate(self._keys):
            v = state_dict[k].detach().clone()
            if v.is_floating_point():
                v = v.to(dtype)
            if k.endswith(".weight") or k.endswith(".bias"):
                p = torch.nn.Parameter(v)
                self.register_parameter(f"p{i}", p)
                self._state[k] = p
            else:
                self.register_buffer(f"b{i}", v)
                self._state[k] = getattr(self, f"b{i}")
        self._dtype = dtype

    def forward(self, x):
        return hourglass_ref.forward(self._state, x.to(self._dtype), training=self.training, update_running_stats=True)

    def state_dict(self, *a, **k):   # upstream key names (what `MannequinChallengeModel.save` writes)
        return {k: v.detach().clone() for k, v in self._state.items()}


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    return m


def _stub_modules(dtype):
    from PIL import Image
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_UNCHANGED, cv2.INTER_AREA, cv2.INTER_NEAREST, cv2.INTER_CUBIC, cv2.INTER_LINEAR = -1, 3, 0, 2, 1

    def imread(path, flags=None):
        with Image.open(path) as im:
            return np.asarray(im)
    cv2.imread = imread
    cv2.imwrite = lambda *a, **k: True
    cv2.applyColorMap = lambda img, cm: np.zeros(np.asarray(img).shape[:2] + (3,), np.uint8)
    wget = types.ModuleType("wget")

    def _no_download(*a, **k):
        raise RuntimeError("no network: the checkpoint must be in ./checkpoints")
    wget.download = _no_download
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def add_histogram(self, *a, **k): pass
    tb.SummaryWriter = SummaryWriter
    tv, tvu = _pkg("torchvision"), types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda x, **k: x[0]
    tv.utils = tvu

    mc, mcm, mco = _pkg("monodepth.mannequin_challenge"), _pkg("monodepth.mannequin_challenge.models"), _pkg("monodepth.mannequin_challenge.options")
    p2p = types.ModuleType("monodepth.mannequin_challenge.models.pix2pix_model")

    class Pix2PixModel:
        def __init__(self, opt):
            sd = self.load_network(None, "G", "best_depth_Ours_Bilinear_inc_3")    # overridden by the reference (:37-38)
            self.netG = _RefNet(sd, dtype)

        def load_network(self, network, network_label, epoch_label):
            raise NotImplementedError

        def switch_to_train(self):
            self.netG.train()

        def switch_to_eval(self):
            self.netG.eval()
    p2p.Pix2PixModel = Pix2PixModel
    mcm.pix2pix_model = p2p
    topt = types.ModuleType("monodepth.mannequin_challenge.options.train_options")

    class TrainOptions:
        def initialize(self):
            self.parser = argparse.ArgumentParser()
            self.parser.add_argument("--input", default="single_view")
    topt.TrainOptions = TrainOptions
    mco.train_options = topt
    mc.models, mc.options = mcm, mco

    def dummy(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        return m
    stubs = {
        "cv2": cv2, "wget": wget, "torch.utils.tensorboard": tb, "torchvision": tv, "torchvision.utils": tvu,
        "monodepth.mannequin_challenge": mc, "monodepth.mannequin_challenge.models": mcm,
        "monodepth.mannequin_challenge.models.pix2pix_model": p2p, "monodepth.mannequin_challenge.options": mco,
        "monodepth.mannequin_challenge.options.train_options": topt,
        "monodepth.midas_v2": _pkg("monodepth.midas_v2"),
        "monodepth.midas_v2.midas_net": dummy("monodepth.midas_v2.midas_net", MidasNet=object),
        "monodepth.monodepth2": _pkg("monodepth.monodepth2"), "monodepth.monodepth2.networks": _pkg("monodepth.monodepth2.networks"),
        "monodepth.monodepth2.networks.resnet_encoder": dummy("monodepth.monodepth2.networks.resnet_encoder", ResnetEncoder=object),
        "monodepth.monodepth2.networks.depth_decoder": dummy("monodepth.monodepth2.networks.depth_decoder", DepthDecoder=object),
    }
    return stubs


@contextlib.contextmanager
def reference_modules(dtype=torch.float32):
    """The reference importable as top-level packages, with the stub set, for the duration of the block."""
    def ours(k):
        return k in _REF_TOP or any(k.startswith(t + ".") for t in _REF_TOP)
    stubs = _stub_modules(dtype)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if ours(k) or k in stubs}
    sys.path.insert(0, REF)
    sys.modules.update(stubs)
    old_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    old_default = torch.get_default_dtype()
    try:
        import loaders.video_dataset as vd
        vd._dtype = dtype
        if dtype == torch.float64:
            torch.set_default_dtype(torch.float64)    # pixel_grid's linspace (utils/geometry.py:15-16) in fp64 too
        yield
    finally:
        torch.set_default_dtype(old_default)
        torch.Tensor.cuda = old_cuda
        for k in [k for k in sys.modules if ours(k) or k in stubs]:
            del sys.modules[k]
        sys.modules.update(saved)
        sys.path.remove(REF)


_LINE = re.compile(r"^Epoch = (\d+), pairs = (\[.*\]), loss = (?:tensor\()?([^,)\s]+)")   # `loss[0]` prints as tensor(v, ...)


def run(clip_path, range_dir, frames, init_state, workdir, *, dtype=torch.float32, num_epochs=2, batch_size=4,
        lambda_r=1.0, lambda_b=0.1, lr=4e-4, seed=0, val_epoch_freq=1, save_epoch_freq=1, do_save_depth=True):
    """Returns {"out_dir", "steps": [(epoch, [[i, j], ...], loss)], "log"}; the artefacts are on disk under out_dir."""
    import json
    os.makedirs(os.path.join(workdir, "checkpoints"), exist_ok=True)
    torch.save({k: v.detach().clone() for k, v in init_state.items()}, os.path.join(workdir, "checkpoints", "mc.pth"))
    cwd = os.getcwd()
    buf = io.StringIO()
    with reference_modules(dtype):
        os.chdir(workdir)
        try:
            import depth_fine_tuning as dft
            parser = dft.DepthFineTuningParams.add_arguments(argparse.ArgumentParser())
            params = parser.parse_args(["--lambda_view_baseline", str(lambda_b), "--lambda_reprojection", str(lambda_r),
                                        "--learning_rate", str(lr), "--batch_size", str(batch_size), "--num_epochs", str(num_epochs),
                                        "--val_epoch_freq", str(val_epoch_freq), "--save_epoch_freq", str(save_epoch_freq)])
            params.path, params.model_type = clip_path, "mc"
            torch.manual_seed(seed)
            torch.set_printoptions(precision=17)       # the per-step loss is only available through the loop's print
            with contextlib.redirect_stdout(buf):
                ft = dft.DepthFineTuner(range_dir, frames, params)
                ft.fine_tune(writer=None)
                if do_save_depth:
                    ft.save_depth()
            out_dir = ft.out_dir
            flow_indices = [list(p) for p in dft.VideoDataset(clip_path, os.path.join(range_dir, "metadata_scaled.npz")).flow_indices]
        finally:
            torch.set_printoptions(profile="default")
            os.chdir(cwd)
    steps = []
    for line in buf.getvalue().splitlines():
        m = _LINE.match(line.strip())
        if m:
            steps.append((int(m.group(1)), json.loads(m.group(2)), float(m.group(3))))
    return {"out_dir": out_dir, "steps": steps, "flow_indices": flow_indices, "log": buf.getvalue()}
