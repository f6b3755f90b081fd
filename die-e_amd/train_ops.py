"""The tower convolutions of the learn loop's training step on the engine's own MFMA kernels.

AlphaZero::train (src/alphazero/alphazero.rs:202-261) is tch autograd over libtorch convolutions in the reference; on
MI355X the PyTorch-ROCm step spends ~100 us per convolution call in MIOpen (Winograd forward, implicit-GEMM weight
gradient, im2col + GEMM data gradient on these 4x6 boards: profiles/r02_train_step_pytorch_kernel_stats.csv) -- 12-17 ms
per 256-sample step, 2.7 % of the bf16 peak, two thirds of the learn loop.  Here the 38 convolutions of the tower run

    forward   y  = conv3x3(x, W) + b           the inference kernel k_conv3x3 / k_conv3x3_sk, raw output (MODE 3)
    dgrad     dx = conv3x3(dy, W^T flipped)     the same kernel on re-packed weights
    wgrad     dW = im2col(x)^T @ dy             hand-written gather + the framework's bf16 GEMM (hipBLASLt / rocBLAS)

in the NHWC token layout [batch*24, 256] bf16 (fp32 master weights, bf16 operands, fp32 accumulation: ordinary mixed
precision), and BatchNorm in train mode + residual add + ReLU of every ResBlock is one fused pass forward and one
backward (PyTorch's batch-norm kernels take 33 us per pass on a [6144, 256] matrix; these take a 3 MB sweep).  The init
block, the heads, the losses and Adam stay PyTorch.  Gradients are checked against fp32 autograd in tests/test_train_gpu.py.
"""
import ctypes as C

import torch

from . import load_library

import os

_N_PACK = 8 * 144 * 64 * 8          # bf16 elements of one packed 256x256x3x3 convolution
WGRAD_GEMM = os.environ.get("DIEE_WGRAD", "") == "gemm"


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(st, what):
    if st != 0:
        raise RuntimeError(f"{what} failed with diee status {st}")


# The bias gradient of a convolution is the column sum of its output gradient, and the batch-norm backward that WRITES that
# gradient (dx) sums its columns on the way out.  Hand-over: the entry keeps a strong reference to dx itself -- while it
# exists the caching allocator cannot give dx's address to another tensor, so "same data_ptr, same shape, same version
# counter" means "this very dx, unmodified" -- and EVERY convolution backward clears it, whether it matched or not.
_dx_colsum = {"dx": None, "colsum": None, "version": None}


def _take_colsum(dy):
    e = dict(_dx_colsum)
    _dx_colsum["dx"] = _dx_colsum["colsum"] = _dx_colsum["version"] = None
    dx = e["dx"]
    if dx is None or dx.data_ptr() != dy.data_ptr() or dx.shape != dy.shape or dx.dtype != dy.dtype or dx._version != e["version"]:
        return None
    return e["colsum"]


class Conv3x3Tok(torch.autograd.Function):
    """y[M,256] = conv3x3 over 4x6 boards of x[M,256] (M = boards*24, bf16) with w[256,256,3,3] (fp32) + b[256] (fp32)"""

    @staticmethod
    def forward(ctx, x, w, b, packed=None):
        """packed: [2, 589824] bf16 (forward, transposed) from pack_tower_weights, or None = pack here"""
        L = load_library()
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[1] == 256 and x.shape[0] % 24 == 0
        assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (256, 256, 3, 3)
        boards = x.shape[0] // 24
        if packed is None:
            packed = torch.empty(2, _N_PACK, dtype=torch.bfloat16, device=x.device)
            _chk(L.diee_train_pack_conv3x3(_ptr(w), _ptr(packed[0]), 0, _stream()), "pack")
            _chk(L.diee_train_pack_conv3x3(_ptr(w), _ptr(packed[1]), 1, _stream()), "pack (transposed)")
        assert packed.dtype == torch.bfloat16 and packed.is_contiguous() and tuple(packed.shape) == (2, _N_PACK)
        y = torch.empty_like(x)
        bias = b.detach().float().contiguous()
        _chk(L.diee_train_conv3x3(_ptr(x), _ptr(packed[0]), _ptr(bias), _ptr(y), boards, _stream()), "conv3x3 forward")
        ctx.save_for_backward(x, packed)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = load_library()
        x, packed = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        boards = x.shape[0] // 24
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _chk(L.diee_train_conv3x3(_ptr(dy), _ptr(packed[1]), None, _ptr(dx), boards, _stream()), "conv3x3 dgrad")
        if ctx.needs_input_grad[1]:
            if WGRAD_GEMM:                                       # DIEE_WGRAD=gemm: im2col + the framework's GEMM (the round-2 first version)
                col = torch.empty(x.shape[0], 2304, dtype=torch.bfloat16, device=x.device)
                _chk(L.diee_train_im2col3x3(_ptr(x), _ptr(col), boards, _stream()), "im2col")
                dwf = torch.matmul(col.t(), dy)                  # [2304 = t*256 + c, 256 = n], fp32 accumulation inside
                dw = dwf.float().view(3, 3, 256, 256).permute(3, 2, 0, 1).contiguous()      # -> [n][c][ky][kx]
            else:                                                # hand-written: transposed LDS reads + MFMA, fp32 out
                dw = torch.empty(256, 256, 3, 3, dtype=torch.float32, device=x.device)
                scratch = torch.empty(int(L.diee_train_wgrad_scratch_floats()), dtype=torch.float32, device=x.device)
                _chk(L.diee_train_wgrad3x3(_ptr(x), _ptr(dy), _ptr(dw), boards, _ptr(scratch), _stream()), "wgrad")
        ready = _take_colsum(dy)                                 # always consumed: a stale entry can never meet a later dy
        if ctx.needs_input_grad[2] and ready is not None:
            db = ready                                           # the batch-norm backward that wrote dy summed its columns already
        elif ctx.needs_input_grad[2]:
            db = torch.empty(256, dtype=torch.float32, device=x.device)
            scratch = torch.empty(int(L.diee_train_scratch_floats(x.shape[0])), dtype=torch.float32, device=x.device)
            _chk(L.diee_train_colsum(_ptr(dy), _ptr(db), x.shape[0], _ptr(scratch), _stream()), "colsum")
        return dx, dw, db, None


class BnReluTok(torch.autograd.Function):
    """y = relu(batch_norm_train(x; gamma, beta) [+ res]) over the rows of x[M,256] bf16; updates running_mean / running_var
    in place (momentum, unbiased variance) like torch.nn.BatchNorm2d.train()"""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, momentum, eps):
        L = load_library()
        assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[1] == 256
        M = x.shape[0]
        y = torch.empty_like(x)
        mean = torch.empty(256, dtype=torch.float32, device=x.device); invstd = torch.empty_like(mean)
        scratch = torch.empty(int(L.diee_train_scratch_floats(M)), dtype=torch.float32, device=x.device)
        if res is not None:
            res = res.contiguous()
        _chk(L.diee_train_bn_relu_fwd(_ptr(x), _ptr(res) if res is not None else None, _ptr(gamma), _ptr(beta),
                                      _ptr(running_mean) if running_mean is not None else None,
                                      _ptr(running_var) if running_var is not None else None, float(momentum), float(eps),
                                      _ptr(mean), _ptr(invstd), _ptr(y), M, _ptr(scratch), _stream()), "bn_relu forward")
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        L = load_library()
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        M = x.shape[0]
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty(256, dtype=torch.float32, device=x.device); dbeta = torch.empty_like(dgamma)
        colsum = torch.empty_like(dgamma)
        scratch = torch.empty(int(L.diee_train_scratch_floats(M)), dtype=torch.float32, device=x.device)
        _chk(L.diee_train_bn_relu_bwd(_ptr(dy), _ptr(y), _ptr(x), _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta),
                                      _ptr(dx), _ptr(dres) if dres is not None else None, _ptr(colsum), M, _ptr(scratch), _stream()),
             "bn_relu backward")
        _dx_colsum["dx"], _dx_colsum["colsum"], _dx_colsum["version"] = dx, colsum, dx._version   # for the convolution in front (next in autograd order)
        return dx, dgamma, dbeta, dres, None, None, None, None


def bn_relu_tok(bn, x, res=None):
    return BnReluTok.apply(x, bn.weight, bn.bias, res, bn.running_mean if bn.training else None,
                           bn.running_var if bn.training else None, bn.momentum, bn.eps)


def conv3x3_tok(x, conv, packed=None):
    return Conv3x3Tok.apply(x, conv.weight, conv.bias, packed)


def pack_tower_weights(convs):
    """the forward and the transposed fragment packing of every 256x256x3x3 convolution in `convs`, one launch:
    [len(convs), 2, 589824] bf16"""
    L = load_library()
    ws = [c.weight for c in convs]
    for w in ws:
        assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (256, 256, 3, 3)
    out = torch.empty(len(ws), 2, _N_PACK, dtype=torch.bfloat16, device=ws[0].device)
    ptrs = (C.c_void_p * len(ws))(*[w.data_ptr() for w in ws])
    _chk(L.diee_train_pack_conv3x3_multi(ptrs, len(ws), _ptr(out), _stream()), "pack (all layers)")
    return out


def to_tokens(h):
    """[B,256,4,6] -> [B*24,256] bf16 (row = board*24 + 6*y + x)"""
    return h.permute(0, 2, 3, 1).reshape(-1, h.shape[1]).to(torch.bfloat16).contiguous()


def from_tokens(t, batch):
    """[B*24,256] -> [B,256,4,6] fp32"""
    return t.view(batch, 4, 6, t.shape[1]).permute(0, 3, 1, 2).float().contiguous()


def _bn_tok(bn, t):
    """BatchNorm2d over (batch, 4, 6) == batch norm of the token matrix over its rows; statistics in fp32"""
    return torch.nn.functional.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)


def forward_train_tokens(net, x):
    """ResNet::forward_t(.., train) (nnet.rs:120-148) with the tower on the engine's kernels; `net` is alphazero.make_resnet()"""
    B = x.shape[0]
    h = torch.relu(net.init_bn(net.init_conv(x)))
    t = to_tokens(h)
    packed = pack_tower_weights([c for blk in net.blocks for c in (blk.conv1, blk.conv2)])
    for i, blk in enumerate(net.blocks):                         # ResBlock::forward_t, nnet.rs:24-34
        g = bn_relu_tok(blk.bn1, conv3x3_tok(t, blk.conv1, packed[2 * i]))
        t = bn_relu_tok(blk.bn2, conv3x3_tok(g, blk.conv2, packed[2 * i + 1]), res=t)
    h = from_tokens(t, B)
    logits = net.p_fc(torch.relu(net.p_bn(net.p_conv(h))).flatten(1))
    value = torch.tanh(net.v_fc(torch.relu(net.v_bn(net.v_conv(h))).flatten(1)))
    return logits, value
