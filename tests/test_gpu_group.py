"""The multi-GPU entry points of the C ABI (rt_group_*), on this box's one GPU: a group may name the
same GPU several times, which exercises everything but the link between two cards -- one context,
stream and host thread per entry, the strip copies into the first context's accumulator, the
rank-ordered sum of the sample shares, the single present pass over the assembled image."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from helpers import assert_bits_equal, copy_cam
from ray_tracing_amd import App, Group, RtError, default_scene, terrain_scene
from ray_tracing_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_ctx,height", [(3, 90), (2, 54), (5, 37)])
def test_tile_mode_is_byte_identical_to_one_context(n_ctx, height):
    """Interleaved 8-row strips over the contexts, copied device to device into the first context's
    binding 7: the chained progressive render (2 frames, then 3 more) equals one context's bit for bit,
    accumulator and presented image, also when the last strip is short (heights 54 and 37)."""
    scene, cam = terrain_scene(160, height, n=48, depth=10, threads=4, leaf_max=4)
    with App.create(copy_cam(cam), scene) as one:
        one.render_frames(2)
        one.render_frames(3)
        want_acc, want_shown = one.read_acc(), one.read_present()
    with Group(copy_cam(cam), scene, [0] * n_ctx) as g:
        assert len(g) == n_ctx
        g.render_frames(2, Group.TILES)
        g.render_frames(3, Group.TILES)
        assert int(g.cam_data.frame) == 5 and float(g.cam_data.frames_without_move) == 5.0
        assert_bits_equal(g.read_acc(), want_acc, "assembled accumulator")
        assert_bits_equal(g.read_present(), want_shown, "presented image of the assembled accumulator")
        rays = [g.context_stats(i)["rays"] for i in range(n_ctx)]
        assert all(r > 0 for r in rays)  # every context traced its strips


def test_sample_mode_sum_equals_the_oracle_shares_added_in_rank_order():
    """Samples split by index over three contexts with independent seeds; the accumulators are added on
    the first GPU in rank order.  Bit-equal to the oracle's shares (rank g: frames g, g + 3, ...)
    added in that order, after one call of 7 samples and after a continuing call of 5 more."""
    scene, cam = default_scene(96, 54)
    world = 3

    def oracle_sum(total_frames):
        total = None
        for rank in range(world):
            c = copy_cam(cam)
            c.frame = rank
            count = len(range(rank, total_frames, world))
            r = O.OracleRenderer(scene, c, trig=O.TRIG_DET, threads=4, frame_stride=world, independent_seeds=True)
            share = r.render(count).acc if count else np.zeros((54, 96, 4), np.float32)
            total = share.copy() if total is None else total + share
        return total

    with Group(copy_cam(cam), scene, [0] * world) as g:
        g.render_frames(7, Group.SAMPLES)
        got7 = g.read_acc()
        assert_bits_equal(got7, oracle_sum(7), "sum of the shares after 7 samples")
        assert np.all(got7[..., 3] == 7.0)  # alpha counts the samples
        shown = g.read_present()
        want = np.power(np.maximum(got7[..., :3] / np.float32(7.0), 0), np.float32(1 / 2.2))
        assert np.allclose(shown[..., :3], want, rtol=1e-5, atol=1e-6)
        g.render_frames(5, Group.SAMPLES)  # frames 7..11 continue the three shares
        assert_bits_equal(g.read_acc(), oracle_sum(12), "sum of the shares after 7 + 5 samples")
        # a camera that moved starts a new sum; fewer samples than contexts leaves shares empty
        g.cam_data = copy_cam(cam)
        g.render_frames(2, Group.SAMPLES)
        assert_bits_equal(g.read_acc(), oracle_sum(2), "a new sum of two samples over three contexts")


def test_group_refuses_what_it_cannot_do():
    lib = L.load()
    scene, cam = default_scene(32, 18)
    g = C.c_void_p()
    assert lib.rt_group_create((C.c_int * 17)(*([0] * 17)), 17, C.byref(g)) == L.RT_ERR_LIMIT
    assert lib.rt_group_create((C.c_int * 1)(99), 1, C.byref(g)) != L.RT_OK and b"99" in lib.rt_group_last_error(None)
    assert lib.rt_group_create(None, 0, C.byref(g)) == L.RT_ERR_INVALID
    with Group(copy_cam(cam), None, [0, 0]) as grp:
        with pytest.raises(RtError) as e:
            grp.render_frames(1)  # no scene on any context
        assert e.value.code == L.RT_ERR_NO_SCENE and "context 0" in str(e.value)
        grp.upload_scene(scene)
        for opt in (L.RT_OPT_FRAME_STRIDE, L.RT_OPT_SEED_MODE, L.RT_OPT_COUNTERS):
            with pytest.raises(RtError):
                grp.set_option(opt, 1)
        with pytest.raises(RtError):
            grp._check(lib.rt_group_render_frames(grp._g, C.byref(grp.cam_data), 1, 7))
        grp.render_frames(1)
        with pytest.raises(RtError):
            grp._check(lib.rt_group_read_acc(grp._g, np.empty(4, np.float32).ctypes.data_as(C.c_void_p), 1))
