"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python -B oracle/make_goldens.py            # rewrites tests/golden/

The reference modules are imported from where they lie; nothing is copied.  Two import shims are
needed: an empty parent package `models` (the real `__init__` star-imports a
name that does not exist), and stand-ins for the handful of `timm` names the DeiT file imports for
registry / init / DropPath -- none of which carries MRLA arithmetic.

Inputs and weights come from `oracle/detgen.py` (bit-reproducible without torch's RNG), so the fixtures
hold only what the reference computed: outputs, gradients, and for the large cases strided samples.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import detgen  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------
# reference import shims
# ------------------------------------------------------------------------------------------------
def import_reference_resnet():
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "resnet", "models")]
    sys.modules["models"] = pkg
    names = ("models.modules.mrla_light_module", "models.modules.mrla_base_module",
             "models.resnet_mrla_light", "models.resnet_mrla_base", "models.utils.drop")
    return {n.rsplit(".", 1)[-1]: importlib.import_module(n) for n in names}


def import_reference_deit():
    from torch import nn

    def mk(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):          # identity for p == 0 / eval, which is all the goldens use
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert self.p == 0.0 or not self.training
            return x

    mk("timm")
    mk("timm.models")
    mk("timm.models.vision_transformer", default_cfgs={}, _cfg=lambda **kw: dict(kw))
    mk("timm.models.registry", register_model=lambda fn: fn)
    mk("timm.models.layers", trunc_normal_=nn.init.trunc_normal_, DropPath=DropPath)
    mk("timm.models.layers.helpers", to_2tuple=lambda v: v if isinstance(v, tuple) else (v, v))
    sys.path.insert(0, os.path.join(REF, "deit"))
    return importlib.import_module("deit_mrla_light"), importlib.import_module("deit_mrla_base")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def N(t):
    return t.detach().cpu().numpy().copy()


def load_det(module, salt=0):
    vals = detgen.fill_state_dict(module.state_dict(), salt)
    module.load_state_dict({k: T(v) for k, v in vals.items()})
    return vals


# ------------------------------------------------------------------------------------------------
# (i) light layer / module / block tail  -- rows a1, a2, a3, a8
# ------------------------------------------------------------------------------------------------
LIGHT_CASES = [  # name, b, c, h, w, d
    ("s64", 2, 64, 8, 8, 32),
    ("s256", 2, 256, 7, 5, 32),
    ("s2048", 1, 2048, 7, 7, 32),
    ("s128d16", 3, 128, 5, 6, 16),
]


def light_inputs(name, b, c, h, w):
    s = detgen.seed_of("light/" + name)
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.25 * detgen.normalish((b, c, h, w), s + 5)
    o = detgen.normalish((b, c, h, w), s + 1)
    gup = detgen.normalish((b, c, h, w), s + 2)
    return x.astype(np.float32), o, gup


def thin(out, step=16):
    """Every `step`-th element (memory order) of the activation-sized tensors; parameter gradients, running statistics and
    masks stay whole (the float64 fixtures stay small)."""
    return {k: (v.ravel()[::step].copy() if v.ndim >= 3 and "grad/" not in k else v) for k, v in out.items()}


def gen_light(ref, f64=False):
    """f64=True: the same reference modules run in DOUBLE precision -> light_blocks_f64.npz (parameter gradients in
    full, every 16th element of the big tensors).  The numpy oracle must agree with these to ~1e-12, which separates
    the oracle's algebra from the reference's own fp32 rounding (1e-7 ... 1e-5 on the cancelling Wq / Wk sums)."""
    L = ref["resnet_mrla_light"]
    dt = torch.float64 if f64 else torch.float32
    out = {}
    for name, b, c, h, w, d in LIGHT_CASES:
        x_np, o_np, g_np = light_inputs(name, b, c, h, w)
        for mode in ("train", "eval", "traindp"):
            p = 0.25 if mode == "traindp" else 0.0
            L.mrla_module.dim_perhead = d
            blk = L.MRLA_Bottleneck(c, c // 4, drop_path=p)
            L.mrla_module.dim_perhead = 32
            load_det(blk, salt=1)
            blk.to(dt)
            blk.train(mode != "eval")
            x = T(x_np).to(dt).requires_grad_(True)
            o = T(o_np).to(dt).requires_grad_(True)
            mask = None
            if p > 0:
                torch.manual_seed(1234)
                mask = torch.floor((1 - p) + torch.rand((b, 1, 1, 1))).to(dt)
                torch.manual_seed(1234)
            layer_out = blk.mrla.mrla(x)                       # a1
            m = blk.mrla(x, o)                                 # a2 (recomputes a1 inside)
            if p > 0:
                torch.manual_seed(1234)
            # (float64 run: the stochastic-depth draw is an INPUT, not arithmetic -- utils/drop.py:20 draws it in x's dtype, and
            # a float64 draw from the same seed is another mask; hand it the float32 draw so both fixtures drop the same images)
            real_rand = torch.rand
            if f64:
                torch.rand = lambda *a, **k: real_rand(*a, **{**k, "dtype": torch.float32}).to(k.get("dtype", torch.float32))
            try:
                y = x + blk.drop_path(blk.bn_mrla(m))          # a3, resnet_mrla_light.py:116
            finally:
                torch.rand = real_rand
            (y * T(g_np).to(dt)).sum().backward()
            k = f"{name}/{mode}/"
            if mode == "train":
                out[k + "layer_out"] = N(layer_out)
                out[k + "m"] = N(m)
            out[k + "out"] = N(y)
            out[k + "dx"] = N(x.grad)
            out[k + "do"] = N(o.grad)
            for pn, pv in blk.named_parameters():
                if pn.startswith(("mrla.", "bn_mrla.")):
                    out[k + "grad/" + pn] = N(pv.grad)
            out[k + "running_mean"] = N(blk.bn_mrla.running_mean)
            out[k + "running_var"] = N(blk.bn_mrla.running_var)
            if mask is not None:
                out[k + "dp_mask"] = N(mask).reshape(b)
    if f64:
        out = thin(out)
        np.savez_compressed(os.path.join(OUT, "light_blocks_f64.npz"), **out)
        print("light_blocks_f64.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")
        return
    # keep the c=2048 case small: strided samples of the big tensors
    for k in list(out):
        if k.startswith("s2048/") and out[k].ndim == 4:
            out[k] = out[k][:, ::8].copy()
    np.savez_compressed(os.path.join(OUT, "light_blocks.npz"), **out)
    print("light_blocks.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


# ------------------------------------------------------------------------------------------------
# (ii) base chains  -- rows a4, a5, a8
# ------------------------------------------------------------------------------------------------
BASE_CASES = [  # name, b, c, h, w, d, T
    ("chain5", 2, 64, 6, 5, 16, 5),
    ("chain23", 1, 32, 2, 3, 16, 23),
    ("chain3cw", 2, 16, 4, 4, 1, 3),
    ("chain23n", 1, 64, 3, 4, 16, 23),      # c % 64 == 0: served by the slot-major NHWC rings as well (BASELINE config 5's path)
]


def base_inputs(name, t, b, c, h, w):
    s = detgen.seed_of(f"base/{name}/{t}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.25 * detgen.normalish((b, c, h, w), s + 5)
    return x.astype(np.float32), detgen.normalish((b, c, h, w), s + 2)


def gen_base(ref, f64=False):
    """f64=True: the chains in double precision -> base_chains_f64.npz (see gen_light)."""
    Bm = ref["resnet_mrla_base"]
    dt = torch.float64 if f64 else torch.float32
    out = {}
    for name, b, c, h, w, d, Tn in BASE_CASES:
        for mode in ("train", "eval"):
            blks = [Bm.MRLA_Bottleneck(c, c // 4, init_cell=(t == 0), channel_wise_mrla=(d == 1))
                    for t in range(Tn)]
            xs, loss, K, V = [], 0.0, None, None
            for t, blk in enumerate(blks):
                load_det(blk, salt=10 + t)
                blk.to(dt)
                blk.train(mode == "train")
                x_np, g_np = base_inputs(name, t, b, c, h, w)
                x = T(x_np).to(dt).requires_grad_(True)
                xs.append(x)
                attn, K, V = blk.mrla(x, K, V)                                  # a4
                y = x + blk.drop_path(blk.relu(blk.bn_mrla(attn)))               # a5, resnet_mrla_base.py:124-127
                loss = loss + (y * T(g_np).to(dt)).sum()
                k = f"{name}/{mode}/{t}/"
                out[k + "attn"] = N(attn)
                out[k + "out"] = N(y)
            loss.backward()
            out[f"{name}/{mode}/K"] = N(K)
            out[f"{name}/{mode}/V"] = N(V)
            for t, blk in enumerate(blks):
                k = f"{name}/{mode}/{t}/"
                out[k + "dx"] = N(xs[t].grad)
                for pn, pv in blk.named_parameters():
                    if pn.startswith(("mrla.", "bn_mrla.")):
                        out[k + "grad/" + pn] = N(pv.grad)
                out[k + "running_mean"] = N(blk.bn_mrla.running_mean)
                out[k + "running_var"] = N(blk.bn_mrla.running_var)
    if f64:
        out = thin(out, 8)
        np.savez_compressed(os.path.join(OUT, "base_chains_f64.npz"), **out)
        print("base_chains_f64.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")
        return
    np.savez_compressed(os.path.join(OUT, "base_chains.npz"), **out)
    print("base_chains.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


# ------------------------------------------------------------------------------------------------
# (iii) DeiT token module  -- rows a6, a7, a8
# ------------------------------------------------------------------------------------------------
TOKEN_CASES = [("t17", 2, 17, 32, 16), ("t197", 2, 197, 192, 16),
               ("t197s", 1, 197, 384, 16), ("t197b", 1, 197, 768, 16)]      # deit_mrlal_small / base widths
GELU_LAYER_CASES = [("g64", 2, 64, 6, 5, 16), ("g192", 2, 192, 14, 14, 16)]     # mrlal_layer on a [b,c,h,w] map


def token_inputs(name, b, n, c):
    s = detgen.seed_of("tok/" + name)
    return (detgen.normalish((b, n, c), s) * 1.5 + 0.3, detgen.normalish((b, n, c), s + 1) * 0.7 - 0.2,
            detgen.normalish((b, n, c), s + 2))


def gen_tokens(deit_light, f64=False):
    """f64=True: the token module / GELU layer in double precision -> token_modules_f64.npz (see gen_light)."""
    dt = torch.float64 if f64 else torch.float32
    out = {}
    for name, b, n, c, d in TOKEN_CASES:
        mod = deit_light.mrlal_module(c, d)
        load_det(mod, salt=3)
        mod.to(dt)
        x_np, o_np, g_np = token_inputs(name, b, n, c)
        x = T(x_np).to(dt).requires_grad_(True)
        o = T(o_np).to(dt).requires_grad_(True)
        y = mod(x, o)                                           # a7 (deit_mrla_light.py:194-209)
        blk_out = x + y                                         # deit_mrla_light.py:234
        (blk_out * T(g_np).to(dt)).sum().backward()
        k = name + "/"
        out[k + "module_out"] = N(y)
        out[k + "dx"] = N(x.grad)
        out[k + "do"] = N(o.grad)
        for pn, pv in mod.named_parameters():
            out[k + "grad/" + pn] = N(pv.grad)
    if not f64:
        for k in list(out):
            if k.startswith("t197") and out[k].ndim == 3:
                out[k] = out[k][:, ::4].copy()
    # the stand-alone map form of the layer (deit_mrla_light.py:157-180, GELU on V), forward and gradients
    for name, b, c, h, w, d in GELU_LAYER_CASES:
        lay = deit_light.mrlal_layer(c, dim_perhead=d)
        load_det(lay, salt=4)
        lay.to(dt)
        s_ = detgen.seed_of("gelu/" + name)
        x = T(detgen.normalish((b, c, h, w), s_)).to(dt).requires_grad_(True)
        y = lay(x)
        (y * T(detgen.normalish((b, c, h, w), s_ + 1)).to(dt)).sum().backward()
        k = name + "/"
        sub = (lambda a: a) if f64 else (lambda a: a[:, ::4]) if c > 64 else (lambda a: a)
        out[k + "out"] = sub(N(y)).copy()
        out[k + "dx"] = sub(N(x.grad)).copy()
        for pn, pv in lay.named_parameters():
            out[k + "grad/" + pn] = N(pv.grad)
    if f64:
        out = thin(out)
        np.savez_compressed(os.path.join(OUT, "token_modules_f64.npz"), **out)
        print("token_modules_f64.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")
        return
    np.savez_compressed(os.path.join(OUT, "token_modules.npz"), **out)
    print("token_modules.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


def import_mmdet_backbone():
    """mmdetection/mmdet/models/backbones/resnet_mrlal.py with mmcv / mmdet stubbed: BaseModule -> nn.Module carrying
    init_cfg, BACKBONES.register_module() -> identity.  No arithmetic lives in the stubs."""
    import importlib
    import types
    root = os.path.join(REF, "mmdetection", "mmdet")

    class BaseModule(torch.nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    nothing = lambda *a, **k: None
    mod("mmcv")
    mod("mmcv.cnn", build_conv_layer=nothing, build_norm_layer=nothing, build_plugin_layer=nothing,
        constant_init=nothing, kaiming_init=nothing)
    mod("mmcv.runner", load_checkpoint=nothing, BaseModule=BaseModule)
    for name, path in (("mmdet", root), ("mmdet.models", os.path.join(root, "models")),
                       ("mmdet.models.backbones", os.path.join(root, "models", "backbones"))):
        mod(name).__path__ = [path]
    mod("mmdet.utils", get_root_logger=nothing)
    mod("mmdet.models.builder", BACKBONES=_Registry())
    return importlib.import_module("mmdet.models.backbones.resnet_mrlal")


DET_SHAPE = (2, 3, 96, 160)


def det_inputs():
    x = detgen.normalish(DET_SHAPE, detgen.seed_of("det/img")).astype(np.float32)
    gs = [detgen.normalish((2, c, 96 // s, 160 // s), detgen.seed_of(f"det/g{c}")).astype(np.float32)
          for c, s in ((256, 4), (512, 8), (1024, 16), (2048, 32))]
    return x, gs


def gen_det():
    """Detection backbone: the four maps in eval mode, and one norm_eval / frozen_stages=1 training step."""
    R = import_mmdet_backbone()
    out = {}
    net = R.ResNet_mrlal(frozen_stages=1, norm_eval=True)
    load_det(net)
    x_np, gs = det_inputs()
    net.eval()
    with torch.no_grad():
        maps = net(T(x_np))
    for i, m in enumerate(maps):
        out[f"eval/map{i}"] = N(m)[:, ::8]                       # every 8th channel
        out[f"eval/map{i}_sum"] = np.array([float(m.double().sum()), float(m.double().abs().sum())])
    net.train()
    maps = net(T(x_np))
    loss = sum((m * T(g)).mean() for m, g in zip(maps, gs))
    loss.backward()
    out["train/loss"] = np.array([loss.item()])
    for i, m in enumerate(maps):
        out[f"train/map{i}_sum"] = np.array([float(m.double().sum()), float(m.double().abs().sum())])
    frozen = []
    for k, p in net.named_parameters():
        if p.grad is None:
            frozen.append(k)
        else:
            g = p.grad.double()
            out["train/gsum/" + k] = np.array([float(g.sum()), float(g.abs().sum())])
    out["train/frozen"] = np.array(sorted(frozen))
    out["state_keys"] = np.array(sorted(net.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "det_backbone.npz"), **out)
    print("det_backbone.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB", len(frozen), "frozen tensors")


def gen_token_base(deit_base):
    """deit_mrla_base.py mrlab_module chain of 5 (history reset at index 4), x_t + module(x_t) as in Block.forward."""
    out = {}
    b, n, c, d = 2, 17, 32, 16
    mods, xs, loss, K, V = [], [], 0.0, None, None
    for t in range(5):
        m = deit_base.mrlab_module(c, d, init_cell=(t % 4 == 0))
        load_det(m, salt=30 + t)
        s_ = detgen.seed_of(f"tokbase/{t}")
        x = T(detgen.normalish((b, n, c), s_) * 1.2 + 0.1).requires_grad_(True)
        y, K, V = m(x, K, V)
        loss = loss + ((x + y) * T(detgen.normalish((b, n, c), s_ + 1))).sum()
        out[f"{t}/module_out"] = N(y)
        mods.append(m); xs.append(x)
    loss.backward()
    for t in range(5):
        out[f"{t}/dx"] = N(xs[t].grad)
        for pn, pv in mods[t].named_parameters():
            out[f"{t}/grad/{pn}"] = N(pv.grad)
    net = deit_base.deit_mrlab_tiny_patch16_224()
    load_det(net)
    net.eval()
    with torch.no_grad():
        out["deit_mrlab_tiny/eval2/logits"] = N(net(T(image_batch(2))))
    np.savez_compressed(os.path.join(OUT, "token_base.npz"), **out)
    print("token_base.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


# ------------------------------------------------------------------------------------------------
# (iv) full models
# ------------------------------------------------------------------------------------------------
def image_batch(b, tag="img"):
    return detgen.normalish((b, 3, 224, 224), detgen.seed_of(tag))


def gen_models(ref, deit_light):
    out = {}
    torch.set_num_threads(8)
    # C1 plumbing config: resnet50_mrlal stands in for the non-existent resnet18_mrlal
    net = ref["resnet_mrla_light"].resnet50_mrlal()
    load_det(net)
    net.eval()
    with torch.no_grad():
        out["resnet50_mrlal/eval8/logits"] = N(net(T(image_batch(8))))
    net.train()
    xb = T(image_batch(4, "img-train"))
    logits = net(xb)
    tgt = torch.arange(4) * 37 % 1000
    loss = torch.nn.functional.cross_entropy(logits, tgt)
    loss.backward()
    out["resnet50_mrlal/train4/logits"] = N(logits)
    out["resnet50_mrlal/train4/loss"] = N(loss)
    for pn, pv in net.named_parameters():
        if ".mrla." in pn or "bn_mrla" in pn or pn in ("conv1.weight", "fc.weight"):
            g = N(pv.grad).ravel()
            out["resnet50_mrlal/train4/gsum/" + pn] = np.array([g.sum(), np.abs(g).sum()], dtype=np.float64)
    out["resnet50_mrlal/train4/rm/layer1.0.bn_mrla"] = N(net.layer1[0].bn_mrla.running_mean)
    out["resnet50_mrlal/train4/rv/layer4.2.bn_mrla"] = N(net.layer4[2].bn_mrla.running_var)
    del net

    net = ref["resnet_mrla_base"].resnet50_mrlab()
    load_det(net)
    net.eval()
    with torch.no_grad():
        out["resnet50_mrlab/eval4/logits"] = N(net(T(image_batch(4))))
    net.train()
    logits = net(xb)
    loss = torch.nn.functional.cross_entropy(logits, tgt)
    loss.backward()
    out["resnet50_mrlab/train4/logits"] = N(logits)
    out["resnet50_mrlab/train4/loss"] = N(loss)
    for pn, pv in net.named_parameters():
        if ".mrla." in pn or "bn_mrla" in pn or pn in ("conv1.0.weight", "fc.weight"):
            g = N(pv.grad).ravel()
            out["resnet50_mrlab/train4/gsum/" + pn] = np.array([g.sum(), np.abs(g).sum()], dtype=np.float64)
    del net

    net = deit_light.deit_mrlal_tiny_patch16_224()
    load_det(net)
    net.eval()
    with torch.no_grad():
        out["deit_mrlal_tiny/eval4/logits"] = N(net(T(image_batch(4))))
    net.train()
    logits = net(xb)
    loss = torch.nn.functional.cross_entropy(logits, tgt)
    loss.backward()
    out["deit_mrlal_tiny/train4/logits"] = N(logits)
    for pn, pv in net.named_parameters():
        if ".mrla." in pn or pn in ("head.weight", "pos_embed"):
            g = N(pv.grad).ravel()
            out["deit_mrlal_tiny/train4/gsum/" + pn] = np.array([g.sum(), np.abs(g).sum()], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "models.npz"), **out)
    print("models.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB")


def gen_state_dict_layout(ref, deit_light, deit_base):
    """Names and shapes of every state_dict entry of the reference models: the checkpoint-compatibility contract
    (resnet/train.py:226-245,333-338 save / resume plain state_dicts)."""
    import json
    layout = {}
    for name, factory in (("resnet50_mrlal", ref["resnet_mrla_light"].resnet50_mrlal),
                          ("resnet101_mrlal", ref["resnet_mrla_light"].resnet101_mrlal),
                          ("resnet50_mrlab", ref["resnet_mrla_base"].resnet50_mrlab),
                          ("resnet101_mrlab", ref["resnet_mrla_base"].resnet101_mrlab),
                          ("deit_mrlal_tiny_patch16_224", deit_light.deit_mrlal_tiny_patch16_224),
                          ("deit_mrlal_small_patch16_224", deit_light.deit_mrlal_small_patch16_224),
                          ("deit_mrlab_tiny_patch16_224", deit_base.deit_mrlab_tiny_patch16_224),
                          # the optional channel-attention modules (off in every BASELINE config): key names only
                          ("resnet50_mrlal+SE+ECA", lambda: ref["resnet_mrla_light"].resnet50_mrlal(SE=True, ECA=[3, 5, 5, 7]))):
        layout[name] = {k: list(v.shape) for k, v in factory().state_dict().items()}
    with open(os.path.join(OUT, "state_dict_layout.json"), "w") as f:
        json.dump(layout, f, separators=(",", ":"))
    print("state_dict_layout.json", {k: len(v) for k, v in layout.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = import_reference_resnet()
    deit_light, deit_base = import_reference_deit()
    which = sys.argv[1:] or ["light", "base", "tokens", "models", "layout", "tokbase", "det", "f64"]
    if "f64" in which:                  # the reference in double precision: pins the oracle's algebra to ~1e-12
        gen_light(ref, f64=True)
        gen_base(ref, f64=True)
        gen_tokens(deit_light, f64=True)
    if "tokbase" in which:
        gen_token_base(deit_base)
    if "det" in which:
        gen_det()
    if "layout" in which:
        gen_state_dict_layout(ref, deit_light, deit_base)
    if "light" in which:
        gen_light(ref)
    if "base" in which:
        gen_base(ref)
    if "tokens" in which:
        gen_tokens(deit_light)
    if "models" in which:
        gen_models(ref, deit_light)


if __name__ == "__main__":
    main()
