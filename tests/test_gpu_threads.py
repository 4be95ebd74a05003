"""Contexts are independent: two host threads, each driving its own context on the same GPU at the
same time (uploads, renders, read-backs interleaved as the scheduler likes), get the same bits as
the same work done one after the other.  The reference drives one context from one thread
(main.rs:102-149); a host with one context per GPU drives several."""
import threading

import numpy as np
import pytest

from ray_tracing_amd import App, RT_OPT_PROBE_HITS, default_scene
from ray_tracing_amd import _lib as L
from ray_tracing_amd import scene as S

pytestmark = pytest.mark.gpu


def _jobs():
    a_scene, a_cam = default_scene(200, 120)
    verts, faces = S.terrain(n=60)
    ident = S.affine_from_srt((1, 1, 1), (0, 0, 0, 1), (0, 0, 0))
    b_scene = S.SceneBuilder().add_mesh(verts, faces, [ident], S.bvh_params(leaf_max=4)).build()
    b_cam = S.make_cam(240, 100, S.look_at((60.0, 45.0, -70.0), (0.0, 0.0, 0.0)), fov=70.0, depth=6)
    return [(a_scene, a_cam), (b_scene, b_cam)]


def _work(scene, cam, rounds, out, key, barrier=None):
    try:
        results = []
        for r in range(rounds):
            if barrier:
                barrier.wait(timeout=60)
            with App.create(L.CamData.from_buffer_copy(cam), scene) as app:
                app.set_option(RT_OPT_PROBE_HITS, 1)
                for _ in range(3):
                    app.render_frames(1 + r)
                tri, inst, tb = app.read_hits()
                results.append((app.read_acc().copy(), tri.copy(), tb.copy()))
        out[key] = results
    except BaseException as e:  # noqa: BLE001 -- handed to the main thread
        out[key] = e


def test_two_threads_with_a_context_each_get_what_one_thread_gets():
    jobs, rounds = _jobs(), 3
    alone = {}
    for k, (scene, cam) in enumerate(jobs):
        _work(scene, cam, rounds, alone, k)
    together, barrier = {}, threading.Barrier(len(jobs))
    threads = [threading.Thread(target=_work, args=(scene, cam, rounds, together, k, barrier))
               for k, (scene, cam) in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=200)
        assert not t.is_alive()
    for k in range(len(jobs)):
        for res in (alone[k], together[k]):
            if isinstance(res, BaseException):
                raise res
        for (acc_a, tri_a, t_a), (acc_t, tri_t, t_t) in zip(alone[k], together[k]):
            assert np.array_equal(acc_a.view(np.uint32), acc_t.view(np.uint32))
            assert np.array_equal(tri_a, tri_t) and np.array_equal(t_a, t_t)
