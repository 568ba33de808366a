"""The reference's lighting A/B switch (shaders/config.h:50-52 USE_NEXT_EVENT_ESTIMATION) in the ORACLE: brute-force path
tracing (liboracle_nee0.so) and next-event estimation with MIS (liboracle.so) are two estimators of the same image. That they
agree within their noise is the one check of light sampling + MIS weights that depends on no restatement being right: a
wrong pdf, a wrong weight or a light counted twice shows as a bias between the two.  (-m gpu: tests/test_gpu_nee.py does the
same on the HIP path at sample counts a CPU cannot afford.)"""
import numpy as np

from conftest import load_app


def two_halves(render, read, n):
    """Running means over iterations [0, n) and [n, 2n) (the second from the means over [0, n) and [0, 2n)) and over [0, 2n)."""
    for it in range(n):
        render(it)
    m1 = read().astype(np.float64)
    for it in range(n, 2 * n):
        render(it)
    m2 = read().astype(np.float64)
    return m1, 2.0 * m2 - m1, m2


def block_means(img, blocks):
    h, w = img.shape[:2]
    bh, bw = h // blocks, w // blocks
    return img[:bh * blocks, :bw * blocks, :3].reshape(blocks, bh, blocks, bw, 3).mean(axis=(1, 3))


def agreement(on, off, blocks):
    """on / off = (first half, second half, whole). Returns (R, relative RMSE on|off, floor on, floor off, relative bias):
    R = sum over blocks of (mean_on - mean_off)^2 / its variance estimated from the half differences; ~1 when unbiased."""
    d = block_means(on[2], blocks) - block_means(off[2], blocks)
    s2 = ((block_means(on[0], blocks) - block_means(on[1], blocks)) ** 2 + (block_means(off[0], blocks) - block_means(off[1], blocks)) ** 2) / 4.0
    scale = on[2][..., :3].mean()
    rel = lambda a, b: float(np.sqrt(((a[..., :3] - b[..., :3]) ** 2).mean()) / scale)
    return (float((d ** 2).sum() / s2.sum()), rel(on[2], off[2]), rel(on[0], on[1]), rel(off[0], off[1]),
            float((on[2][..., :3].mean() - off[2][..., :3].mean()) / scale))


def test_oracle_nee_on_and_off_estimate_the_same_image(twk, orc):
    """C1 (Lambert Cornell box, 1x1 area light) at 32x32; path length 2..48 so that the truncated tail — which the two
    estimators cut differently: NEE's last vertex still gets its light sample — is below 1e-4 of the image."""
    app = load_app(twk, "system_rtigo3_cornell_box_c1.txt", "scene_rtigo3_cornell_box_c1.txt", (32, 32))
    st = app.state
    st.pathLengths[0], st.pathLengths[1] = 2, 48
    results = []
    for nee, n in ((True, 48), (False, 768)):
        ref = orc.Oracle(miss=app.info.miss, nee=nee)
        ref.loadApplication(app, state=st)
        results.append(two_halves(lambda it: ref.render(it, threads=8), ref.getOutputBufferHost, n))
        ref.close()
    R, cross, floor_on, floor_off, bias = agreement(results[0], results[1], blocks=4)
    print(f"oracle NEE on (2x48 spp) vs off (2x768 spp), 32x32: R {R:.3f}, relative RMSE cross {cross:.4f}, floors on {floor_on:.4f} off {floor_off:.4f}, relative bias of the image mean {bias:+.5f}")
    assert results[1][2][..., :3].max() > 0.5, "the brute-force image found the light"
    assert R < 3.0, "NEE-on and NEE-off block means differ by more than their noise"
    assert abs(bias) < 0.02
