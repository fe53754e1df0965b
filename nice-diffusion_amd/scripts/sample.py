#!/usr/bin/env python3
"""Sample images from a diffusion model on an AMD Instinct GPU.

Command-line surface of the reference's ``scripts/sample.py`` (same flags, preset-by-filename rule, x_T from the CPU
generator after ``torch.manual_seed``, uint8 conversion, file naming).  Differences: the denoising loop and the
uint8/HWC conversion run as HIP kernels; ``--cpu`` is refused (no CPU path in this build); ``--start_img`` needs
an image reader (PIL or cv2) and ``--upsample`` needs basicsr, both imported lazily.
"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from nicediffusion import _hip  # noqa: E402
from nicediffusion.utils import make_argparser, get_dicts_from_args  # noqa: E402
from nicediffusion.model import DiffusionModel  # noqa: E402
from nicediffusion.diffusion import Diffusion  # noqa: E402


def split_argv(argv):
    """A single quoted argument string is accepted too: every argv entry is re-split on spaces."""
    out = []
    for a in argv:
        out.extend(a.split(' '))
    return out


def to_uint8_hwc(x, invert=False):
    """[B,C,R,R] fp32 in [-1,1] on the GPU -> uint8 [B,R,R,C] on the host: ((x+1)*127.5).clamp(0,255), truncated."""
    lib = _hip.load()
    B, C, R, _ = x.shape
    st = torch.cuda.current_stream().cuda_stream
    nhwc = torch.empty(B * R * R * C, dtype=torch.float32, device=x.device)
    _hip.check(lib.nd_nchw_to_nhwc(x.contiguous().data_ptr(), nhwc.data_ptr(), B, C, R * R, C, st), 'nd_nchw_to_nhwc')
    u8 = torch.empty(B, R, R, C, dtype=torch.uint8, device=x.device)
    _hip.check(lib.nd_to_uint8_hwc(nhwc.data_ptr(), C, u8.data_ptr(), B, R * R, C, 1 if invert else 0, st),
               'nd_to_uint8_hwc')
    return u8.cpu().numpy()


def saved_bytes(out, in_channels):
    """uint8 image bytes exactly as the reference saves / shows them (scripts/sample.py:94-100,164,170-171).
    Colour: trunc(v), v = clamp((x+1)*127.5, 0, 255).  One-channel models: the reference inverts in float, truncates,
    then inverts the bytes again, 255 - trunc(255 - v) -- which is ceil(v) for non-integer v, NOT trunc(v)."""
    if in_channels == 1:
        return 255 - to_uint8_hwc(out, invert=True)
    return to_uint8_hwc(out)


def resize_linear_u8(img, width, height):
    """cv2.resize(img, dsize=(width, height)) with its default INTER_LINEAR on uint8 HWC data, restated (OpenCV
    imgproc resize.cpp): half-pixel centres, NO antialiasing when shrinking, 11-bit fixed-point coefficients, and the
    exact-2x shrink special case (2x2 box average).  Used when cv2 itself is not installed."""
    img = np.ascontiguousarray(img)
    sh, sw = img.shape[:2]
    if sw == 2 * width and sh == 2 * height:          # INTER_LINEAR -> INTER_AREA fast path
        a = img.astype(np.int32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)

    def taps(dst, src):
        scale = src / dst
        f = (np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5
        i0 = np.floor(f).astype(np.int64)
        frac = (f - i0).astype(np.float32)
        lo = i0 < 0
        i0[lo], frac[lo] = 0, 0.0
        hi = i0 >= src - 1
        i0[hi], frac[hi] = src - 1, 0.0
        i1 = np.minimum(i0 + 1, src - 1)
        w1 = np.rint(frac * 2048.0).astype(np.int32)
        w0 = np.rint((1.0 - frac) * 2048.0).astype(np.int32)
        return i0, i1, w0, w1
    x0, x1, a0, a1 = taps(width, sw)
    y0, y1, b0, b1 = taps(height, sh)
    src = img.astype(np.int32)
    rows = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]          # horizontal pass, x2048
    r0, r1 = rows[y0], rows[y1]
    out = ((((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2)
    return np.clip(out, 0, 255).astype(np.uint8)


def load_start_image(path, resolution):
    """scripts/sample.py:55-58: imread (BGR) -> resize (INTER_LINEAR) -> /127.5 - 1 in float64 -> RGB, CHW, fp32."""
    try:
        from cv2 import imread, resize
        img = resize(imread(path), dsize=(resolution, resolution))[..., ::-1]
    except ImportError:
        from PIL import Image
        img = resize_linear_u8(np.asarray(Image.open(path).convert('RGB')), resolution, resolution)
    arr = np.ascontiguousarray(img).astype(np.float64) / 127.5 - 1
    return torch.from_numpy(arr).permute(2, 0, 1).float()


def main(argv=None):
    argv = split_argv(sys.argv[1:] if argv is None else argv)
    args = make_argparser('diff_sample').parse_args(argv)
    other, model_args, diff_args = get_dicts_from_args(args)
    if other['cpu']:
        raise _hip.NdHipError('--cpu: this build samples on an AMD GPU only (the CPU path is the reference itself)')
    if not torch.cuda.is_available():
        raise _hip.NdHipError('no GPU visible')
    device = torch.device('cuda')
    if other['seed'] is not None:
        torch.manual_seed(other['seed'])
    wordy, n_batches, B = other['wordy'], other['num_samples'], other['batch_size']
    labels_arg, save_path = other['labels'], other['save_path']
    conditional = model_args['num_classes'] is not None
    C, R = model_args['in_channels'], model_args['resolution']

    model = DiffusionModel(**model_args)
    model.load_state_dict(torch.load(other['model_path'], map_location='cpu'), strict=True)
    model.to(device).eval()
    if wordy:
        print('Model made from {} with {} parameters! :)'.format(other['model_path'],
                                                                 sum(p.numel() for p in model.parameters())))
        print('Starting Diffusion! There are {} samples of {} images each'.format(n_batches, B))
    diffusion = Diffusion(model=model, **diff_args, device=device)

    start = None
    if other['start_img'] is not None and other['steps_to_do'] is not None:
        img = load_start_image(other['start_img'], R)
        start = img[None].repeat(B, 1, 1, 1).to(device)
    if conditional and len(labels_arg) != 0:
        assert len(labels_arg) == n_batches, 'please provide NUM_SAMPLES={} labels'.format(n_batches)

    results = []
    t0 = time.time()
    for i in range(n_batches):
        if start is None:
            data = torch.randn([B, C, R, R]).to(device)          # CPU generator, then copied (reference behaviour)
            steps = diff_args['rescaled_num_steps']
        else:
            steps = other['steps_to_do'] * diff_args['rescaled_num_steps'] // diff_args['original_num_steps']
            data = diffusion.diffuse(x_0=start, steps_to_do=steps)
        if conditional:
            if len(labels_arg) == 0:
                labels = torch.randint(low=0, high=model_args['num_classes'], size=(B,), device=device)
            else:
                labels = torch.full(size=(B,), fill_value=labels_arg[i], device=device)
        else:
            labels = None
        if wordy:
            print('Denoising sample {}! :)'.format(i + 1))
        out = diffusion.denoise(x=data, kwargs={'y': labels}, batch_size=B, progress=wordy, steps_to_do=steps)
        results.append((saved_bytes(out, C), None if labels is None else labels.cpu().numpy()))
    torch.cuda.synchronize()
    if wordy:
        dt = time.time() - t0
        print('{} images in {:.2f} s = {:.3f} images/sec'.format(n_batches * B, dt, n_batches * B / dt))
    if other['upsample']:
        raise NotImplementedError('--upsample (Real-ESRGAN) is a third-party post-processing net outside this build')

    if save_path is None:
        from nicediffusion.utils import imshow
        import matplotlib.pyplot as plt
        for imgs, labels in results:
            for b in range(B):
                plt.close('all')
                im = imgs[b] if C != 1 else imgs[b, ..., 0]
                imshow(im, title='Output Image' if labels is None else 'Output Image, Label={}'.format(labels[b]))
                plt.waitforbuttonpress()
    else:
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        counts = {}
        for imgs, labels in results:
            for b in range(B):
                if labels is not None:
                    lab = int(labels[b])
                    name = '{}_sample{}.jpg'.format(lab, counts.get(lab, 0))
                    counts[lab] = counts.get(lab, 0) + 1
                else:
                    name = 'sample{}.jpg'.format(counts.get(None, 0))
                    counts[None] = counts.get(None, 0) + 1
                im = imgs[b] if C != 1 else imgs[b, ..., 0]
                plt.imsave(save_path + name, im)          # plain concatenation, as the reference does
    if wordy:
        print('Done! have a nice day')


if __name__ == '__main__':
    main()
