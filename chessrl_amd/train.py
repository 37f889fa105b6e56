"""Training step of the tower -- PyTorch-ROCm, fp32.

Mirror of what the reference does between two self-play games
(/root/reference/src/chessrl/selfplay.py:98-108 -> agent.py:64-89 -> model.py:69-72,83-99):
``Adam(lr=0.002)`` on ``categorical_crossentropy(policy) + mean_squared_error(value)`` plus the
``kernel_regularizer='l2'`` (= ``l2(0.01)``) term of every conv / dense kernel (model.py:33-58,
113-118), BatchNormalization in training mode (batch statistics, eps 1e-3, momentum 0.99), one
batch = the augmented positions of ``batch_size`` games.

Keras semantics restated here (TensorFlow is absent in this image: *from recollection*; the
tests hold this module to an independently written CPU restatement of the same step):
  * crossentropy on probabilities: ``p / sum(p)``, clipped to [1e-7, 1 - 1e-7], ``-sum(t * log p)``,
    mean over the batch;
  * BN moving statistics: ``moving = 0.99 * moving + 0.01 * batch``, the moving variance from the
    UNBIASED batch variance (TF fused batch norm), normalisation by the biased one;
  * Adam (TF2 ``optimizer_v2``): ``lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)``,
    ``w -= lr_t * m / (sqrt(v) + 1e-7)`` -- epsilon is NOT bias-corrected, unlike ``torch.optim.Adam``,
    so the update is written out here.
The trained weights go back into the self-play path through ``ChessModel.load_dict`` (BN folded,
fp16 tiles repacked for the fused HIP trunk).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import BN_EPS, IN_PLANES, N_POLICY, PAD_PLANES

L2 = 0.01                 # keras.regularizers.l2 default, what the string 'l2' resolves to
BN_MOMENTUM = 0.99        # keras.layers.BatchNormalization default
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 0.002, 0.9, 0.999, 1e-7
CCE_EPS = 1e-7            # keras.backend.epsilon()


class _Im2Col3x3(torch.autograd.Function):
    """Patch matrix of a 3x3 'same' convolution on NHWC boards and its adjoint: the two hand-written
    HIP kernels of csrc/train_ops.hpp (one launch each, any batch size, deterministic backward)."""

    @staticmethod
    def _run(fn, src, dst, n_boards, channels):
        import ctypes
        from . import _lib
        rc = fn(ctypes.c_void_p(torch.cuda.current_stream(src.device).cuda_stream),
                ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), n_boards, channels)
        if rc != 0:
            raise _lib.HipLibraryError("train_ops kernel failed (%d)" % rc)

    @staticmethod
    def forward(ctx, x):
        from . import _lib
        x = x.contiguous()
        b, c = x.shape[0], x.shape[3]
        ctx.shape = (b, c)
        cols = torch.empty((b * 64, 9 * c), dtype=torch.float32, device=x.device)
        _Im2Col3x3._run(_lib.lib().crl_im2col3x3_f32, x, cols, b, c)
        return cols

    @staticmethod
    def backward(ctx, gcols):
        from . import _lib
        b, c = ctx.shape
        gcols = gcols.contiguous()
        gx = torch.empty((b, 8, 8, c), dtype=torch.float32, device=gcols.device)
        _Im2Col3x3._run(_lib.lib().crl_col2im3x3_f32, gcols, gx, b, c)
        return gx


class _PatchGemm(torch.autograd.Function):
    """out = cols @ wm + bias with a split-K weight gradient.

    The weight gradient cols^T @ gout has K = B*64 (~24 000 for one game) against a 1152 x 128 output:
    as ONE GEMM rocBLAS runs it at 42 TFLOP/s (too few output tiles for 256 CUs).  Splitting the rows
    into SPLIT independent chunks (a batched GEMM) and adding the partial products fills the chip.
    The summation order is fixed, so the step stays deterministic."""

    SPLIT = 16                                              # divides B*64 for every B

    @staticmethod
    def forward(ctx, cols, wm, bias):
        ctx.save_for_backward(cols, wm)
        return torch.addmm(bias, cols, wm)

    @staticmethod
    def backward(ctx, gout):
        cols, wm = ctx.saved_tensors
        gout = gout.contiguous()
        n, s = cols.shape[0], _PatchGemm.SPLIT
        gcols = gout @ wm.t() if ctx.needs_input_grad[0] else None
        gw = torch.bmm(cols.view(s, n // s, -1).transpose(1, 2), gout.view(s, n // s, -1)).sum(dim=0)
        return gcols, gw, gout.sum(dim=0)


def im2col3x3(x):
    """x [B,8,8,C] fp32 -> [B*64, 9*C], columns ordered (ky, kx, c).  CUDA tensors go through the HIP
    kernels; the torch expression below is the same map for CPU tensors (unit tests of the module
    on the host -- ``Trainer`` itself refuses to run without a GPU)."""
    if x.is_cuda:
        return _Im2Col3x3.apply(x)
    b, c = x.shape[0], x.shape[3]
    xp = F.pad(x, (0, 0, 1, 1, 1, 1))
    return torch.cat([xp[:, dy:dy + 8, dx:dx + 8, :] for dy in range(3) for dx in range(3)],
                     dim=-1).reshape(b * 64, 9 * c)


class TrainTower(nn.Module):
    """The tower with its BatchNorm layers unfolded, fp32, NHWC activations."""

    def __init__(self, blocks, filters):
        super().__init__()
        self.blocks_n, self.filters = blocks, filters

        def bn(c):
            return nn.BatchNorm2d(c, eps=BN_EPS, momentum=1.0 - BN_MOMENTUM)
        self.stem = nn.Conv2d(PAD_PLANES, filters, 3, padding=1)
        self.conv1 = nn.ModuleList([nn.Conv2d(filters, filters, 3, padding=1) for _ in range(blocks)])
        self.bn1 = nn.ModuleList([bn(filters) for _ in range(blocks)])
        self.conv2 = nn.ModuleList([nn.Conv2d(filters, filters, 3, padding=1) for _ in range(blocks)])
        self.bn2 = nn.ModuleList([bn(filters) for _ in range(blocks)])
        self.policy_conv, self.policy_bn = nn.Conv2d(filters, 2, 1), bn(2)
        self.policy_fc = nn.Linear(128, N_POLICY)
        self.value_conv, self.value_bn = nn.Conv2d(filters, 1, 1), bn(1)
        self.value_fc1 = nn.Linear(64, 256)
        self.value_fc2 = nn.Linear(256, 1)

    # ---- Keras-layout dict <-> parameters ----------------------------------------------------
    def _convs(self):
        yield "stem", self.stem, None
        for i in range(self.blocks_n):
            yield "block%d.conv1" % i, self.conv1[i], ("block%d.bn1" % i, self.bn1[i])
            yield "block%d.conv2" % i, self.conv2[i], ("block%d.bn2" % i, self.bn2[i])
        yield "policy.conv", self.policy_conv, ("policy.bn", self.policy_bn)
        yield "value.conv", self.value_conv, ("value.bn", self.value_bn)

    def _denses(self):
        yield "policy.dense", self.policy_fc
        yield "value.dense1", self.value_fc1
        yield "value.dense2", self.value_fc2

    @torch.no_grad()
    def load_keras_dict(self, w):
        def t(a):
            return torch.from_numpy(np.asarray(a, np.float32))
        for name, conv, bn in self._convs():
            k = t(w[name + ".kernel"]).permute(3, 2, 0, 1)                # HWIO -> OIHW
            if name == "stem":
                kp = torch.zeros(k.shape[0], PAD_PLANES, 3, 3)
                kp[:, :IN_PLANES] = k
                k = kp
            conv.weight.copy_(k)
            conv.bias.copy_(t(w[name + ".bias"]))
            if bn is not None:
                bname, m = bn
                m.weight.copy_(t(w[bname + ".gamma"]))
                m.bias.copy_(t(w[bname + ".beta"]))
                m.running_mean.copy_(t(w[bname + ".mean"]))
                m.running_var.copy_(t(w[bname + ".var"]))
        for name, fc in self._denses():
            fc.weight.copy_(t(w[name + ".kernel"]).t())
            fc.bias.copy_(t(w[name + ".bias"]))

    @torch.no_grad()
    def to_keras_dict(self):
        w = {}
        for name, conv, bn in self._convs():
            k = conv.weight.detach().float().cpu()
            if name == "stem":
                k = k[:, :IN_PLANES]
            w[name + ".kernel"] = k.permute(2, 3, 1, 0).contiguous().numpy()
            w[name + ".bias"] = conv.bias.detach().cpu().numpy().copy()
            if bn is not None:
                bname, m = bn
                w[bname + ".gamma"] = m.weight.detach().cpu().numpy().copy()
                w[bname + ".beta"] = m.bias.detach().cpu().numpy().copy()
                w[bname + ".mean"] = m.running_mean.cpu().numpy().copy()
                w[bname + ".var"] = m.running_var.cpu().numpy().copy()
        for name, fc in self._denses():
            w[name + ".kernel"] = fc.weight.detach().cpu().t().contiguous().numpy()
            w[name + ".bias"] = fc.bias.detach().cpu().numpy().copy()
        w["meta.blocks"] = np.array(self.blocks_n)
        w["meta.filters"] = np.array(self.filters)
        return w

    def regularized(self):
        """Every kernel carrying ``kernel_regularizer='l2'`` (all convs and denses, not biases/BN)."""
        return [c.weight for _, c, _ in self._convs()] + [fc.weight for _, fc in self._denses()]

    # ---- forward ---------------------------------------------------------------------------------
    # Training batches are whole games, so every batch has a different number of positions.  MIOpen
    # looks up / compiles solvers per (batch, shape) for forward, backward-data and backward-weights
    # (measured: ~1.5 s for every new batch size), so the convolutions are written as what they are
    # on an 8x8 board: a [B*64, 9*Cin] x [9*Cin, Cout] GEMM over the NHWC activations (patch matrix and
    # its adjoint: the HIP kernels of csrc/train_ops.hpp), which rocBLAS/hipBLASLt run for any B;
    # autograd gives the two backward GEMMs.  BatchNorm runs over the flattened [B*64, C] rows.
    @staticmethod
    def _conv3x3(x, conv):
        """x [B,8,8,Cin] -> [B,8,8,Cout]; kernel OIHW viewed as [(ky,kx,c), o]."""
        b, cin = x.shape[0], x.shape[3]
        wm = conv.weight.permute(2, 3, 1, 0).reshape(9 * cin, -1)
        return _PatchGemm.apply(im2col3x3(x), wm, conv.bias).view(b, 8, 8, -1)

    @staticmethod
    def _conv1x1(x, conv):
        b = x.shape[0]
        return torch.addmm(conv.bias, x.reshape(b * 64, -1), conv.weight.view(conv.weight.shape[0], -1).t()
                           ).view(b, 8, 8, -1)

    @staticmethod
    def _bn(x, m):
        c = x.shape[-1]
        return F.batch_norm(x.reshape(-1, c), m.running_mean, m.running_var, m.weight, m.bias,
                            m.training, m.momentum, m.eps).view(x.shape)

    def forward(self, planes):
        """planes: [B,8,8,128] NHWC (the encoder kernel's buffer, any float dtype).
        Returns (policy probabilities [B,1968], value [B,1])."""
        x = self._conv3x3(planes.float(), self.stem)         # no BN / activation (model.py:33-34)
        for c1, b1, c2, b2 in zip(self.conv1, self.bn1, self.conv2, self.bn2):
            y = F.relu(self._bn(self._conv3x3(x, c1), b1))
            y = self._bn(self._conv3x3(y, c2), b2)
            x = F.relu(x + y)
        b = x.shape[0]
        p = F.relu(self._bn(self._conv1x1(x, self.policy_conv), self.policy_bn)).reshape(b, 128)
        p = torch.softmax(self.policy_fc(p), dim=-1)         # Keras Flatten order (h, w, c)
        v = F.relu(self._bn(self._conv1x1(x, self.value_conv), self.value_bn)).reshape(b, 64)
        v = torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
        return p, v


def keras_losses(policy, value, move_index, result, regularized):
    """(total, policy crossentropy, value mse, l2 term) as Keras reports them."""
    p = policy / policy.sum(dim=-1, keepdim=True)
    p = p.clamp(CCE_EPS, 1.0 - CCE_EPS)
    cce = -torch.log(p.gather(1, move_index.view(-1, 1))[:, 0]).mean()
    mse = ((value[:, 0] - result) ** 2).mean()
    reg = sum(L2 * (k * k).sum() for k in regularized)
    return cce + mse + reg, cce, mse, reg


class KerasAdam(object):
    """TF2 Keras ``Adam`` (see module docstring), fused over all tensors with ``torch._foreach``."""

    def __init__(self, params, lr=ADAM_LR, beta_1=ADAM_B1, beta_2=ADAM_B2, epsilon=ADAM_EPS):
        self.params = [p for p in params]
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.t = 0
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        self.t += 1
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        # m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
        torch._foreach_mul_(self.m, self.b1)
        torch._foreach_add_(self.m, grads, alpha=1.0 - self.b1)
        torch._foreach_mul_(self.v, self.b2)
        torch._foreach_addcmul_(self.v, grads, grads, value=1.0 - self.b2)
        lr_t = self.lr * float(np.sqrt(1.0 - self.b2 ** self.t)) / (1.0 - self.b1 ** self.t)
        denom = torch._foreach_sqrt(self.v)
        torch._foreach_add_(denom, self.eps)
        torch._foreach_addcdiv_(self.params, self.m, denom, value=-lr_t)


class Trainer(object):
    """Holds the unfolded fp32 tower, the optimizer state and runs ``fit_generator``-style epochs."""

    def __init__(self, weights, device):
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("training needs an MI355X (no CPU fallback in the product path)")
        self.net = TrainTower(int(weights["meta.blocks"]), int(weights["meta.filters"]))
        self.net.load_keras_dict(weights)
        self.net.to(self.device).train()
        self.opt = KerasAdam(self.net.parameters())

    def train_on_batch(self, planes, move_index, result):
        """One optimizer step; returns dict(loss, policy_loss, value_loss, reg_loss, accuracy)."""
        logs = self.backward_on_batch(planes, move_index, result)
        self.opt.step()
        return {k: v.item() for k, v in logs.items()}

    def backward_on_batch(self, planes, move_index, result):
        """Forward + backward of one batch: the gradients are left on the parameters, no step is taken (the
        data-parallel trainer averages them over the ranks first).  Returns the metrics as device scalars."""
        self.net.train()
        self.opt.zero_grad()
        policy, value = self.net(planes)
        total, cce, mse, reg = keras_losses(policy, value, move_index, result, self.net.regularized())
        total.backward()
        acc = (policy.argmax(dim=-1) == move_index).float().mean()
        return {"loss": total.detach(), "policy_out_loss": cce.detach(), "value_out_loss": mse.detach(),
                "reg_loss": reg.detach(), "policy_out_accuracy": acc}

    @torch.no_grad()
    def evaluate(self, planes, move_index, result):
        self.net.eval()
        policy, value = self.net(planes)
        total, cce, mse, reg = keras_losses(policy, value, move_index, result, self.net.regularized())
        self.net.train()
        return {"loss": total.item(), "policy_out_loss": cce.item(), "value_out_loss": mse.item()}

    def fit_generator(self, generator, epochs=1, val_gen=None, verbose=0, log=None):
        """Keras ``fit_generator`` on a ``Sequence``: every batch once per epoch, batch order
        shuffled by the global ``np.random`` (Sequence default ``shuffle=True``)."""
        history = []
        for epoch in range(epochs):
            order = np.random.permutation(len(generator))
            logs = []
            for idx in order:
                planes, move_index, result = generator.device_batch(int(idx), self.device)
                logs.append(self.train_on_batch(planes, move_index, result))
            summary = {k: float(np.mean([l[k] for l in logs])) for k in logs[0]} if logs else {}
            if val_gen is not None and len(val_gen) > 0:
                vl = [self.evaluate(*val_gen.device_batch(i, self.device)) for i in range(len(val_gen))]
                summary.update({"val_" + k: float(np.mean([l[k] for l in vl])) for k in vl[0]})
            summary["epoch"] = epoch
            history.append(summary)
            if log is not None:
                log(summary)
            if verbose:
                print("epoch %d: %s" % (epoch, summary))
        return history

    def weights(self):
        return self.net.to_keras_dict()


def fit_data_parallel(trainer, generator, group=None, epochs=1, log=None):
    """One ``fit_generator`` pass with the ranks of ``group`` training TOGETHER: every rank holds the same
    weights and its own games; step i takes one game per rank, the gradients are averaged over the ranks that
    still have a game (one flat all_reduce over RCCL), and every rank applies the same Adam step.  The
    BatchNorm moving statistics, which every rank updates from its own batches, are averaged at the end.
    NOT the reference's arithmetic: its ``fit_generator`` takes one Adam step per game (model.py:83-99), this
    takes one per ``world`` games -- the price of a trainer whose throughput grows with the number of GPUs.
    Every rank must call it with the same ``epochs``; returns this rank's history."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    cpu_group = dist.get_backend(group) != "nccl"
    dev = trainer.device
    params = trainer.opt.params
    sizes = [p.numel() for p in params]
    history = []
    for epoch in range(epochs):
        order = np.random.permutation(len(generator))
        n = torch.tensor([len(order)], dtype=torch.int64, device="cpu" if cpu_group else dev)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        logs = []
        for i in range(int(n.item())):
            have = i < len(order)
            if have:
                planes, move_index, result = generator.device_batch(int(order[i]), dev)
                have = planes.shape[0] > 0
            if have:
                logs.append(trainer.backward_on_batch(planes, move_index, result))
                grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
            else:
                grads = [torch.zeros_like(p) for p in params]
            flat = torch.cat([g.reshape(-1) for g in grads] + [torch.full((1,), float(have), device=dev)])
            if cpu_group:
                host = flat.cpu()
                dist.all_reduce(host, group=group)
                flat = host.to(dev)
            else:
                dist.all_reduce(flat, group=group)
            flat = flat[:-1] / flat[-1].clamp(min=1.0)
            off = 0
            for p, k in zip(params, sizes):
                p.grad = flat[off:off + k].view_as(p)
                off += k
            trainer.opt.step()
        # BatchNorm moving statistics: the mean over the ranks
        bufs = [b for m in trainer.net.modules() if isinstance(m, nn.BatchNorm2d) for b in (m.running_mean, m.running_var)]
        flatb = torch.cat([b.reshape(-1) for b in bufs])
        if cpu_group:
            host = flatb.cpu()
            dist.all_reduce(host, group=group)
            flatb = host.to(dev)
        else:
            dist.all_reduce(flatb, group=group)
        flatb /= world
        off = 0
        with torch.no_grad():
            for b in bufs:
                b.copy_(flatb[off:off + b.numel()].view_as(b))
                off += b.numel()
        summary = {k: float(np.mean([l[k].item() for l in logs])) for k in logs[0]} if logs else {}
        summary["epoch"] = epoch
        summary["ranks"] = world
        history.append(summary)
        if log is not None:
            log(summary)
    return history


def broadcast_weights(weights, device, src=0):
    """All ranks end up with rank ``src``'s weight dict: ONE flat fp32 broadcast over RCCL (the only
    collective of a play+train round besides the record gather)."""
    import torch.distributed as dist
    names = sorted(k for k in weights if not k.startswith("meta."))
    flat = torch.cat([torch.from_numpy(np.asarray(weights[k], np.float32)).reshape(-1) for k in names])
    flat = flat.to(device) if dist.get_backend() == "nccl" else flat
    dist.broadcast(flat, src=src)
    flat = flat.cpu().numpy()
    out, off = {k: weights[k] for k in weights if k.startswith("meta.")}, 0
    for k in names:
        shape = np.asarray(weights[k]).shape
        n = int(np.prod(shape))
        out[k] = flat[off:off + n].reshape(shape).copy()
        off += n
    return out
