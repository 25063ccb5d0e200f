"""Generates the committed golden vectors from the CPU oracle (oracle/crucible_oracle.c).

The reference (Rust) cannot be built or imported in this image and holds no fixtures for
the render path, so these vectors pin the oracle restatement itself ("parity unpinned"
beyond the reference's arithmetic KATs, see tests/test_oracle_kat.py): any later change
to the oracle or to the toolchain that moves a bit shows up as a diff against them.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from crucible_amd import _abi as A  # noqa: E402
from crucible_amd.demo_builder import book1_end_scene, checkered_spheres  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
import scenes  # noqa: E402

SEED = 0xC0FFEE


def vectors(o):
    """Per-function vectors: inputs are seeded, outputs come from the oracle probes."""
    rs = np.random.RandomState(1234)
    out = {}
    out["rng_u64"] = np.zeros(16, dtype=np.uint64)
    o.lib.oracle_rng_u64(SEED, 5, 7, 16, out["rng_u64"].ctypes.data)
    out["rng_uniform"] = np.zeros(16, dtype=o.np_real)
    o.lib.oracle_rng_uniforms(SEED, 5, 7, 16, o._p(out["rng_uniform"]))
    n = 64
    orig = o.arr(rs.uniform(-3, 3, size=(n, 3)))
    dirs = o.arr(rs.uniform(-1, 1, size=(n, 3)))
    sph = o.arr(np.concatenate([rs.uniform(-1, 1, size=(n, 3)), rs.uniform(0.2, 2.0, size=(n, 1))], axis=1))
    tri = o.arr(rs.uniform(-2, 2, size=(n, 9)))
    box = o.arr(np.sort(rs.uniform(-2, 2, size=(n, 3, 2)), axis=2).reshape(n, 6))
    out.update(orig=orig, dirs=dirs, sph=sph, tri=tri, box=box)
    sh = np.zeros((n, 11), dtype=o.np_real)
    th = np.zeros((n, 11), dtype=o.np_real)
    bh = np.zeros(n, dtype=np.int32)
    for i in range(n):
        sh[i, 0] = o.lib.oracle_sphere_hit(o._p(sph[i]), o._p(orig[i]), o._p(dirs[i]), 0.001, np.inf, o._p(sh[i, 1:]))
        th[i, 0] = o.lib.oracle_triangle_hit(o._p(tri[i]), o._p(orig[i]), o._p(dirs[i]), 0.001, np.inf, o._p(th[i, 1:]))
        bh[i] = o.lib.oracle_aabb_hit(o._p(box[i]), o._p(orig[i]), o._p(dirs[i]), 0.001, np.inf)
    out.update(sphere_hit=sh, triangle_hit=th, aabb_hit=bh)
    # materials / textures / sky / camera through the mixed scene
    sc = scenes.mixed_scene(width=32, samples=2)
    flat = sc.flatten()
    h = o.scene_create(flat)
    try:
        n_mat = flat.desc.n_materials
        rec = o.arr(np.concatenate([rs.uniform(0.1, 5, size=(n, 1)), rs.uniform(-2, 2, size=(n, 3)),
                                    rs.normal(size=(n, 3)), rs.uniform(0, 1, size=(n, 2)), rs.randint(0, 2, size=(n, 1))], axis=1))
        rec[:, 4:7] /= np.linalg.norm(rec[:, 4:7], axis=1, keepdims=True).astype(o.np_real)
        sc_out = np.zeros((n, 11), dtype=o.np_real)
        for i in range(n):
            sc_out[i, 0] = o.lib.oracle_scatter(h, i % n_mat, o._p(orig[i]), o._p(dirs[i]), o._p(rec[i]), SEED, i, 3,
                                                o._p(sc_out[i, 1:]))
        out.update(scatter_rec=rec, scatter_out=sc_out, scatter_n_mat=np.int32(n_mat))
        n_tex = flat.desc.n_textures
        tv = np.zeros((n, 3), dtype=o.np_real)
        for i in range(n):
            o.lib.oracle_texture_value(h, i % n_tex, rec[i, 7], rec[i, 8], o._p(rec[i, 1:4].copy()), o._p(tv[i]))
        sky = np.zeros((n, 3), dtype=o.np_real)
        for i in range(n):
            o.lib.oracle_sky(h, o._p(dirs[i]), o._p(sky[i]))
        out.update(texture_value=tv, sky=sky)
        wh = np.zeros((n, 12), dtype=o.np_real)
        mat = np.zeros(1, dtype=np.int32)
        for i in range(n):
            wh[i, 0] = o.lib.oracle_world_hit(h, o._p(orig[i]), o._p(dirs[i]), 0.0, 0.001, np.inf, o._p(wh[i, 1:11]), mat.ctypes.data)
            wh[i, 11] = mat[0] if wh[i, 0] else -1
        out["world_hit"] = wh
    finally:
        o.scene_destroy(h)
    cam = sc.scene_cam
    cr = np.zeros((n, 8), dtype=o.np_real)
    cd, p = cam.desc(), cam.params(SEED, o.real_type)
    for i in range(n):
        o.lib.oracle_camera_ray(cd, p, i % cam.image_width, (i * 7) % cam.image_height, i % 5, o._p(cr[i]))
    out["camera_ray"] = cr
    return out


def images(o):
    out = {}
    for name, sc in (("book1_64x36_spp4", book1_end_scene(1, scene_seed=1, image_width=64, samples=4)),
                     ("checkered_48x27_spp3", checkered_spheres(1, image_width=48, samples=3)),
                     ("mixed_64x36_spp4", scenes.mixed_scene(64, 4)),
                     ("mixed_anim_48x27_spp4", scenes.mixed_scene(48, 4, animate=True)),
                     ("mixed_nosky_40x22_spp3", scenes.mixed_scene(40, 3, sky=False))):
        img, st = o.render_image(sc, seed=SEED, n_threads=8)
        out[name] = img
        out[name + "_stats"] = np.array([st["segments"], st["node_tests"], st["prim_tests"], st["texel_fetches"]], dtype=np.uint64)
    return out


if __name__ == "__main__":
    for rt, tag in ((A.CR_REAL_F64, "f64"), (A.CR_REAL_F32, "f32")):
        o = Oracle(rt)
        np.savez_compressed(os.path.join(HERE, f"vectors_{tag}.npz"), **vectors(o))
        np.savez_compressed(os.path.join(HERE, f"images_{tag}.npz"), **images(o))
        print("wrote", tag)
