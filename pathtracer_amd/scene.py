"""A flat scene (include/ptx_scene.h) held as numpy arrays.

Scene files are written by the host scene kit (pathtracer_amd/host) or, for
fixtures, by the reference dump tool; this module only reads them and hands
out a ptx_scene_desc whose pointers refer to the numpy buffers.
"""
import ctypes as C
import struct

import numpy as np

from . import abi


class Scene:
    def __init__(self, objects, nodes, lights, images, texels, camera):
        self.objects = np.ascontiguousarray(objects, dtype=abi.OBJECT_DTYPE)
        self.nodes = np.ascontiguousarray(nodes, dtype=abi.NODE_DTYPE)
        self.lights = np.ascontiguousarray(lights, dtype=np.int32)
        self.images = np.ascontiguousarray(images, dtype=abi.IMAGE_DTYPE)
        self.texels = np.ascontiguousarray(texels, dtype=np.uint8)
        self.camera = np.ascontiguousarray(camera, dtype=abi.CAMERA_DTYPE).reshape(1)

    @classmethod
    def from_file(cls, path):
        with open(path, "rb") as f:
            buf = f.read()
        magic, n_obj, n_nodes, n_lights, n_images, n_tex = struct.unpack_from("<QiiiiQ", buf, 0)
        if magic != abi.PTX_SCENE_MAGIC:
            raise ValueError(f"{path}: not a scene file")
        off = 32
        cam = np.frombuffer(buf, abi.CAMERA_DTYPE, 1, off); off += abi.CAMERA_DTYPE.itemsize
        objects = np.frombuffer(buf, abi.OBJECT_DTYPE, n_obj, off); off += n_obj * abi.OBJECT_DTYPE.itemsize
        nodes = np.frombuffer(buf, abi.NODE_DTYPE, n_nodes, off); off += n_nodes * abi.NODE_DTYPE.itemsize
        lights = np.frombuffer(buf, "<i4", n_lights, off); off += 4 * n_lights
        images = np.frombuffer(buf, abi.IMAGE_DTYPE, n_images, off); off += n_images * abi.IMAGE_DTYPE.itemsize
        texels = np.frombuffer(buf, np.uint8, n_tex, off); off += n_tex
        if off != len(buf):
            raise ValueError(f"{path}: trailing or missing bytes")
        return cls(objects.copy(), nodes.copy(), lights.copy(), images.copy(), texels.copy(), cam.copy())

    @classmethod
    def from_npz(cls, path):
        """Scene arrays saved with numpy.savez (how the larger test scenes are stored)."""
        z = np.load(path, allow_pickle=False)
        return cls(z["objects"], z["nodes"], z["lights"], z["images"], z["texels"], z["camera"])

    @classmethod
    def load(cls, path):
        return cls.from_npz(path) if path.endswith(".npz") else cls.from_file(path)

    def to_file(self, path):
        with open(path, "wb") as f:
            f.write(struct.pack("<QiiiiQ", abi.PTX_SCENE_MAGIC, len(self.objects), len(self.nodes), len(self.lights),
                                len(self.images), self.texels.size))
            for a in (self.camera, self.objects, self.nodes, self.lights, self.images, self.texels):
                f.write(a.tobytes())

    def desc(self):
        """ptx_scene_desc pointing into this object's arrays (keep the Scene alive while it is used)."""
        d = abi.SceneDesc()
        d.magic = abi.PTX_SCENE_MAGIC
        d.n_objects, d.n_nodes, d.n_lights, d.n_images = len(self.objects), len(self.nodes), len(self.lights), len(self.images)
        d.n_texel_bytes = self.texels.size
        d.objects = self.objects.ctypes.data
        d.nodes = self.nodes.ctypes.data
        d.lights = self.lights.ctypes.data
        d.images = self.images.ctypes.data
        d.texels = self.texels.ctypes.data
        C.memmove(C.addressof(d.cam), self.camera.ctypes.data, abi.CAMERA_DTYPE.itemsize)
        return d
