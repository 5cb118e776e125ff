"""numpy restatement of the two steps either side of the path that SURVEY.md section 8(f)3 moves on-device.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__``).

``adam_step``       the optimiser the reference builds at semseg.py:106-111 / pcdseg.py:133-138:
                    ``torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-08, weight_decay)``.  The arithmetic
                    lives in a third-party dependency -- PyTorch (2.10.0 in this image; the reference pins nothing,
                    README.md:9-13) ``torch/optim/adam.py::_single_tensor_adam``, the non-capturable, non-amsgrad,
                    ``maximize=False`` branch -- restated here in fp32 numpy with the scalars formed as Python
                    doubles exactly as that function forms them.
``prepare_cloud``   ``SemKITTI_Loader.__getitem__`` (data_utils/SemKITTI_Loader.py:93-113): ``pcd_normalize``
                    (:23-30), ``pcd_jitter`` (:17-21, training only) and the with-replacement resampling
                    (:110-113), drawing from numpy's global generator in the reference's order.

Parity pin: ``tools/make_golden_train.py`` runs ``torch.optim.Adam`` and the reference's own two loader functions
(executed from /root/reference in the development container) and stores inputs + outputs in
``tests/golden/g8_train.npz``; ``tests/test_oracle_golden.py`` checks this file against them: bit-equal for the
loader; for Adam <= 2e-7 relative on the parameters and <= 1e-6 on the moments after 12 steps, as its ATen kernels fuse multiply-adds this restatement does not.
"""
import numpy as np

F32 = np.float32


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """One Adam update in place on fp32 arrays; ``step`` is t >= 1 of this update.  Returns ``param``."""
    beta1, beta2 = betas
    g = grad
    if weight_decay != 0:
        g = g + F32(weight_decay) * param                          # grad.add(param, alpha=weight_decay)
    w = F32(1 - beta1)
    d = g - exp_avg                                                # exp_avg.lerp_(grad, 1 - beta1)
    exp_avg[...] = exp_avg + w * d if w < 0.5 else g - d * (F32(1) - w)
    exp_avg_sq[...] = exp_avg_sq * F32(beta2) + F32(1 - beta2) * g * g
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    denom = np.sqrt(exp_avg_sq) / F32(bias_correction2 ** 0.5) + F32(eps)
    param[...] = param - F32(step_size) * (exp_avg / denom)        # addcdiv_(exp_avg, denom, value=-step_size)
    return param


SCALE = np.array([70, 70, 3], F32)                                 # SemKITTI_Loader.py:25-27


def normalize(pcd):
    """[M,4] raw x,y,z,intensity -> the network's input range (SemKITTI_Loader.py:23-30)."""
    out = np.empty_like(pcd, dtype=F32)
    out[:, :3] = pcd[:, :3].astype(F32) / SCALE
    out[:, 3] = (pcd[:, 3].astype(F32) - F32(0.5)) * F32(2)
    return np.clip(out, F32(-1), F32(1))


def jitter_noise(M, C=4, sigma=0.01, clip=0.05, rng=np.random):
    """The clipped jitter rows of pcd_jitter (SemKITTI_Loader.py:17-19): fp64 draw, clip, cast to fp32."""
    return np.clip(sigma * rng.randn(M, C), -clip, clip).astype(F32)


def prepare_cloud(pcd, label, npoints, train, rng=np.random):
    """One ``__getitem__``: returns (points [npoints,4] fp32, label [npoints], noise or None, choice)."""
    out = normalize(pcd)
    noise = None
    if train:
        noise = jitter_noise(out.shape[0], out.shape[1], rng=rng)
        out = noise + out                                          # jittered_data += pcd (:20)
    choice = rng.choice(out.shape[0], npoints, replace=True)       # :110-111
    return out[choice], label[choice], noise, choice
