"""`App`: the host-side mirror of the reference's `vulkan::App`
(program/src/modules/vulkan.rs:78-236, 374) on top of librt_hip.so.

Same surface -- create / render / update_pos / update_mouse / destroy and a
public `cam_data` -- minus the window: frames stay in HBM and are read back on
request.  Every method goes through the C ABI; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .scene import Scene, _ptr


class App:
    def __init__(self):
        raise TypeError("use App.create(cam_data, scene)")

    @classmethod
    def create(cls, cam_data, scene, device=0, stream=None):
        """App::create (vulkan.rs:124-168): takes the camera and the scene buffers,
        uploads the scene once."""
        self = object.__new__(cls)
        self._lib = L.load()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(device, C.byref(self._ctx))
        if rc != L.RT_OK:
            msg = self._lib.rt_last_error(None).decode()
            self._ctx = None
            raise L.RtError(rc, msg)
        self.device = int(device)
        self.cam_data = cam_data
        self.orientation = (C.c_float * 2)(0.0, 0.0)  # vulkan.rs:84, 161
        self.scene = None
        if stream is not None:
            self._check(self._lib.rt_set_stream(self._ctx, C.c_void_p(stream)))
        if scene is not None:
            self.upload_scene(scene)
        return self

    # ---- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc != L.RT_OK:
            raise L.RtError(rc, self._lib.rt_last_error(self._ctx).decode())

    def destroy(self):
        """App::destroy (vulkan.rs:374)."""
        if getattr(self, "_ctx", None):
            self._lib.rt_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        self.destroy()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    # ---- scene ------------------------------------------------------------
    def upload_scene(self, scene: Scene):
        s = scene
        self._check(self._lib.rt_upload_scene(
            self._ctx, C.byref(s.info), _ptr(s.vertices), len(s.vertices), _ptr(s.triangles), len(s.triangles),
            _ptr(s.objects), len(s.objects), _ptr(s.instances), len(s.instances), _ptr(s.bvh), len(s.bvh)))
        self.scene = scene
        if scene.materials is not None:
            self.set_materials(scene.materials)

    def update_instances(self, instances):
        """New transforms for the uploaded scene's instances (an INSTANCE_DT array of the same
        length): the per-frame refill of the instance buffer, vulkan.rs:238-353."""
        instances = np.ascontiguousarray(instances)
        self._check(self._lib.rt_update_instances(self._ctx, _ptr(instances), len(instances)))

    def set_materials(self, materials):
        if materials is None:
            self._check(self._lib.rt_set_materials(self._ctx, None, 0))
            return
        arr = (L.Material * len(materials))(*materials)
        self._check(self._lib.rt_set_materials(self._ctx, arr, len(materials)))

    def set_option(self, option, value):
        self._check(self._lib.rt_set_option(self._ctx, option, int(value)))

    def set_rows(self, strip_height, rank, world):
        self._check(self._lib.rt_set_rows(self._ctx, strip_height, rank, world))

    # ---- camera: vulkan.rs:95-121 ---------------------------------------------
    def update_pos(self, delta):
        self._lib.rt_cam_update_pos(C.byref(self.cam_data), (C.c_float * 3)(*delta))

    def update_mouse(self, mouse_x, mouse_y):
        self._lib.rt_cam_update_mouse(C.byref(self.cam_data), self.orientation, mouse_x, mouse_y)

    # ---- render: vulkan.rs:171-236 + the bookkeeping of main.rs:147-148 ----------
    def render(self, advance=True):
        """One sample per pixel with the current cam_data.  With `advance` the
        frame counters move on exactly as the reference's event loop moves them."""
        if advance:
            self._check(self._lib.rt_render_frames(self._ctx, C.byref(self.cam_data), 1))
        else:
            self._check(self._lib.rt_render_frame(self._ctx, C.byref(self.cam_data)))

    def render_frames(self, n):
        self._check(self._lib.rt_render_frames(self._ctx, C.byref(self.cam_data), n))

    def render_frames_noalloc(self, n, acc_ptr, present_ptr=None):
        """rt_render_frames_noalloc: n frames into caller-owned device buffers (addresses as
        integers), with no allocation inside the call."""
        self._check(self._lib.rt_render_frames_noalloc(self._ctx, C.byref(self.cam_data), n, C.c_void_p(acc_ptr),
                                                       C.c_void_p(present_ptr) if present_ptr else None))

    def synchronize(self):
        self._check(self._lib.rt_synchronize(self._ctx))

    # ---- read-back ------------------------------------------------------------
    def _n(self):
        return int(self.cam_data.canvas_width) * int(self.cam_data.canvas_height)

    def _shape(self):
        return int(self.cam_data.canvas_height), int(self.cam_data.canvas_width)

    def read_acc(self):
        out = np.empty((*self._shape(), 4), np.float32)
        self._check(self._lib.rt_read_acc(self._ctx, _ptr(out), self._n()))
        return out

    def read_present(self):
        out = np.empty((*self._shape(), 4), np.float32)
        self._check(self._lib.rt_read_present(self._ctx, _ptr(out), self._n()))
        return out

    def read_present_rgba8(self):
        """The displayed image as (H, W, 4) bytes, converted on the GPU (a quarter of the transfer)."""
        out = np.empty((*self._shape(), 4), np.uint8)
        self._check(self._lib.rt_read_present_rgba8(self._ctx, _ptr(out), self._n()))
        return out

    def present_begin(self):
        """Start quantising and copying the displayed image; returns at once (see present_end)."""
        self._check(self._lib.rt_present_begin(self._ctx))

    def present_end(self):
        """Wait for the copy present_begin started; (H, W, 4) bytes (a copy of the library's buffer)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.rt_present_end(self._ctx, C.byref(p), C.byref(n)))
        h, w = self._shape()
        assert n.value == h * w
        buf = (C.c_uint8 * (n.value * 4)).from_address(p.value)
        return np.frombuffer(buf, np.uint8).reshape(h, w, 4).copy()

    # ---- image files, in place of the swapchain copy (vulkan.rs:1045-1101) ------
    def save_png(self, path):
        """The displayed image (binding 8) as an 8-bit PNG."""
        img = self.read_present()
        h, w = self._shape()
        rc = self._lib.rt_save_png(str(path).encode(), _ptr(img), w, h)
        if rc != 0:
            raise L.RtError(rc, f"could not write {path}")

    def save_exr(self, path, samples=None):
        """Mean linear radiance as a float OpenEXR file: the accumulator divided by the number
        of samples in it (default: cam_data.frames_without_move, which render() keeps equal to
        that number, main.rs:147-148)."""
        acc = self.read_acc()
        if samples is None:
            samples = max(1.0, float(self.cam_data.frames_without_move))
        h, w = self._shape()
        rc = self._lib.rt_save_exr(str(path).encode(), _ptr(acc), w, h, 1.0 / float(samples))
        if rc != 0:
            raise L.RtError(rc, f"could not write {path}")

    def write_acc(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        h, w = self._shape()
        self._check(self._lib.rt_write_acc(self._ctx, _ptr(rgba), w * h, w, h))

    def clear_acc(self):
        self._check(self._lib.rt_clear_acc(self._ctx))

    def read_hits(self):
        tri, inst, tb = (np.empty(self._shape(), np.uint32) for _ in range(3))
        self._check(self._lib.rt_read_hits(self._ctx, _ptr(tri), _ptr(inst), _ptr(tb), self._n()))
        return tri, inst, tb

    def read_counters(self):
        box, tt = (np.empty(self._shape(), np.uint32) for _ in range(2))
        self._check(self._lib.rt_read_counters(self._ctx, _ptr(box), _ptr(tt), self._n()))
        return box, tt

    def stats(self):
        s = L.Stats()
        self._check(self._lib.rt_get_stats(self._ctx, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in L.Stats._fields_}

    def sched_stats(self):
        s = L.SchedStats()
        self._check(self._lib.rt_get_sched_stats(self._ctx, C.byref(s)))
        return {"iterations": list(s.iterations), "lanes": list(s.lanes), "cycles": list(s.cycles), "wait_cycles": list(s.wait_cycles),
                "site_wait_cycles": list(s.site_wait_cycles),
                "shade": [int(s.shade_iterations), int(s.shade_slots), int(s.shade_cycles)]}

    def kernel_clock_mhz(self):
        """Shader clock (MHz) the persistent kernels ran at since the last reset_stats, from the
        kernels' own cycle and 100 MHz counters; None when no such launch has run."""
        cyc, ticks = C.c_uint64(), C.c_uint64()
        self._check(self._lib.rt_get_kernel_clock(self._ctx, C.byref(cyc), C.byref(ticks)))
        return cyc.value / ticks.value * 100.0 if ticks.value else None

    def reset_stats(self):
        self._check(self._lib.rt_reset_stats(self._ctx))

    def acc_device_ptr(self):
        p = C.c_void_p()
        h, w = self._shape()
        self._check(self._lib.rt_acc_device_ptr(self._ctx, w, h, C.byref(p)))
        return p.value

    def acc_tensor(self):
        """The accumulation buffer as a torch tensor (H, W, 4) aliasing device memory."""
        import torch

        h, w = self._shape()

        class _Alias:
            __cuda_array_interface__ = {"shape": (h, w, 4), "typestr": "<f4", "data": (self.acc_device_ptr(), False),
                                        "version": 2, "strides": None}

        return torch.as_tensor(_Alias(), device=torch.device("cuda", self.device))  # the context's GPU, whatever torch's current one is

    def last_render_ms(self):
        ms = C.c_float()
        self._check(self._lib.rt_last_render_ms(self._ctx, C.byref(ms)))
        return float(ms.value)


class Group:
    """Several GPUs of one node as one renderer (rt_group_* of the C ABI): one context per entry of
    `devices`, the scene replicated, the assembled image on the first GPU.  The reference has one
    App on one GPU; a group is what its event loop (main.rs:129-149) would own instead."""

    TILES, SAMPLES = L.RT_GROUP_TILES, L.RT_GROUP_SAMPLES

    def __init__(self, cam_data, scene, devices):
        self._lib = L.load()
        self._g = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        rc = self._lib.rt_group_create(devs, len(devices), C.byref(self._g))
        if rc != L.RT_OK:
            msg = self._lib.rt_group_last_error(None).decode()
            self._g = None
            raise L.RtError(rc, msg)
        self.cam_data = cam_data
        self.devices = list(devices)
        if scene is not None:
            self.upload_scene(scene)

    def _check(self, rc):
        if rc != L.RT_OK:
            raise L.RtError(rc, self._lib.rt_group_last_error(self._g).decode())

    def destroy(self):
        if getattr(self, "_g", None):
            self._lib.rt_group_destroy(self._g)
            self._g = None

    __del__ = destroy

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __len__(self):
        return int(self._lib.rt_group_size(self._g))

    def upload_scene(self, scene: Scene):
        s = scene
        self._check(self._lib.rt_group_upload_scene(
            self._g, C.byref(s.info), _ptr(s.vertices), len(s.vertices), _ptr(s.triangles), len(s.triangles),
            _ptr(s.objects), len(s.objects), _ptr(s.instances), len(s.instances), _ptr(s.bvh), len(s.bvh)))
        if scene.materials is not None:
            arr = (L.Material * len(scene.materials))(*scene.materials)
            self._check(self._lib.rt_group_set_materials(self._g, arr, len(scene.materials)))

    def set_option(self, option, value):
        self._check(self._lib.rt_group_set_option(self._g, option, int(value)))

    def context_option(self, i, option, value):
        ctx = self._lib.rt_group_context(self._g, i)
        if not ctx or self._lib.rt_set_option(ctx, option, int(value)) != L.RT_OK:
            raise L.RtError(L.RT_ERR_INVALID, f"option {option} on context {i}")

    def context_stats(self, i):
        s = L.Stats()
        if self._lib.rt_get_stats(self._lib.rt_group_context(self._g, i), C.byref(s)) != L.RT_OK:
            raise L.RtError(L.RT_ERR_HIP, f"statistics of context {i}")
        return {k: int(getattr(s, k)) for k, _ in L.Stats._fields_}

    def render_frames(self, n, mode=L.RT_GROUP_TILES):
        self._check(self._lib.rt_group_render_frames(self._g, C.byref(self.cam_data), n, mode))

    def synchronize(self):
        self._check(self._lib.rt_group_synchronize(self._g))

    def _shape(self):
        return int(self.cam_data.canvas_height), int(self.cam_data.canvas_width)

    def read_acc(self):
        h, w = self._shape()
        out = np.empty((h, w, 4), np.float32)
        self._check(self._lib.rt_group_read_acc(self._g, _ptr(out), h * w))
        return out

    def read_present(self):
        h, w = self._shape()
        out = np.empty((h, w, 4), np.float32)
        self._check(self._lib.rt_group_read_present(self._g, _ptr(out), h * w))
        return out
