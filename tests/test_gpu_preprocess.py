"""vgpa_preprocess_frames (csrc/preprocess.hip, host mirror videogpa_amd/model_utils.py) against PIL's own outputs
(tests/golden/preprocess.npz, made by calling PIL: utils/model_utils.py:16-85) and against the numpy oracle at sizes the fixture
does not hold.  Integer work up to the final /255: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_preprocess_matches_pil_fixture_bit_exact():
    from videogpa_amd.model_utils import preprocess_images_from_numpy
    z = np.load(os.path.join(HERE, "golden", "preprocess.npz"))
    for n in sorted({k.split("__")[0] for k in z.files}):
        frames, want, mode = z[n + "__frames"], z[n + "__expect_u8"], str(z[n + "__mode"])
        got = preprocess_images_from_numpy(frames, mode)
        assert got.shape == (1,) + want.shape and got.dtype == torch.float32
        ref = torch.from_numpy(want).float().div(255.0)      # ToTensor of the uint8 image (:52)
        assert torch.equal(got[0].cpu(), ref), (n, (got[0].cpu() - ref).abs().max().item())


@pytest.mark.parametrize("T,H,W,mode", [(10, 720, 1280, "crop"), (3, 1080, 1920, "pad"), (2, 1280, 720, "crop"), (2, 1280, 720, "pad"), (1, 60, 4001, "crop"),
                                        (2, 33, 17, "pad")])
def test_preprocess_matches_oracle_bit_exact(T, H, W, mode):
    from oracle import preprocess as pp
    from videogpa_amd.model_utils import preprocess_images_from_numpy, preprocessed_size
    rng = np.random.default_rng(H * 7 + W)
    frames = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    frames[0, : H // 2] = 255
    frames[0, :, : W // 3] = 0       # hard edges: ringing beyond [0, 255] must clip after each pass
    got = preprocess_images_from_numpy(frames, mode)
    want = pp.preprocess_images_from_numpy(frames, mode)
    assert tuple(got.shape) == want.shape == (1, T, 3) + preprocessed_size(H, W, mode)
    assert np.array_equal(got.cpu().numpy(), want)


def test_preprocess_errors_follow_the_reference():
    from videogpa_amd.model_utils import preprocess_images_from_numpy, prepare_inputs
    with pytest.raises(ValueError, match="must be"):
        preprocess_images_from_numpy(np.zeros((2, 8, 8), np.uint8))
    with pytest.raises(ValueError, match="crop"):
        preprocess_images_from_numpy(np.zeros((1, 8, 8, 3), np.uint8), "stretch")
    with pytest.raises(ValueError, match="No frames"):
        prepare_inputs(np.zeros((0, 8, 8, 3), np.uint8))
    # a device-resident uint8 clip is taken as is
    f = torch.randint(0, 256, (2, 40, 60, 3), dtype=torch.uint8, device="cuda")
    a = preprocess_images_from_numpy(f)
    b = preprocess_images_from_numpy(f.cpu().numpy())
    assert torch.equal(a, b)


def test_video_processor_runs_the_vggt_wrapper_on_device():
    """VideoProcessor(vggt_model=...): utils/model_utils.py:89-122 around a stand-in network -- the network sees the device-preprocessed
    clip [1, T, 3, h, 518] (equal to the oracle's), its outputs lose the batch axis, world_points is aliased, pose_enc is decoded."""
    from oracle import preprocess as pp
    from videogpa_amd.process_video import VideoProcessor
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, (3, 72, 128, 3), dtype=np.uint8)
    seen = {}

    def net(images):
        seen["images"] = images
        seen["autocast"] = torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16     # utils/model_utils.py:99-102
        T, h, w = images.shape[1], images.shape[3], images.shape[4]
        pe = torch.zeros(1, T, 9, device=images.device)
        pe[..., 6] = 1.0
        pe[..., 7:] = 0.9
        return {"pose_enc": pe, "world_points": torch.ones(1, T, h, w, 3, device=images.device), "depth_conf": torch.ones(1, T, h, w, device=images.device)}

    vp = VideoProcessor(metrics={}, backbone="vggt", vggt_model=net)
    preds = vp.backbone_fn(frames)
    assert np.array_equal(seen["images"].cpu().numpy(), pp.preprocess_images_from_numpy(frames))
    assert preds["images"].shape == (3, 3, 294, 518) and preds["pose_enc"].shape == (3, 9) and preds["depth_conf"].shape == (3, 294, 518)
    assert preds["world_points_from_depth"] is preds["world_points"]
    assert seen["autocast"] and preds["extrinsic"].shape == (3, 3, 4) and preds["intrinsic"].shape == (3, 3, 3)    # decoded before the squeeze (:104-106)

    class Net(torch.nn.Module):                      # a real module is moved to the device and put in eval mode (:92)
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(1))

        def forward(self, images):
            seen["mode"] = (self.training, self.w.device.type)
            return net(images)

    m = Net().train()
    VideoProcessor(metrics={}, backbone="vggt", vggt_model=m).backbone_fn(frames)
    assert seen["mode"] == (False, "cuda")
