"""Synthetic inputs in the reference's binary formats (no downloads possible).

Writers for: .vks scenes (format: reference src/scene.c:419-483, written the way
tools/io_export_vulkan_blender28.py:472-530 quantises), LTC fit files
(src/ltc_table.c:31-82), constant .vkt material textures (src/textures.c:101-172)
and the light / camera set-ups of the BASELINE.json configurations.  Everything
is seeded and deterministic."""
import math
import os
import struct

import numpy as np

# ---- .vks -----------------------------------------------------------------------


def _morton_10(v):
    v = v.astype(np.uint64) & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def encode_octahedral_normals(normals):
    """(..., 3) unit normals -> two uint16 arrays (octahedral map, -1 -> 1, 0 -> 32768, +1 -> 65535)."""
    n = np.asarray(normals, np.float64)
    o = n[..., :2] / np.abs(n).sum(axis=-1, keepdims=True)
    sign = np.where(o >= 0.0, 1.0, -1.0)
    folded = (1.0 - np.abs(o[..., ::-1])) * sign
    o = np.where(n[..., 2:3] <= 0.0, folded, o)
    q = np.floor(o * 32767.0 + 32768.5)
    q = np.clip(q, 0, 65535).astype(np.uint16)
    return q[..., 0], q[..., 1]


def write_vks(path, positions, normals, uvs, material_indices, material_names, sort_triangles=True, shuffle_seed=None):
    """positions, normals: (T, 3, 3); uvs: (T, 3, 2); material_indices: (T,).
    Returns the dict of buffers exactly as they are stored in the file.  shuffle_seed: store the triangles
    in a random order instead of along a Morton curve (neighbours in the file are then unrelated in space:
    the worst case for everything that combines the contributions of consecutive triangles)."""
    positions = np.asarray(positions, np.float32)
    normals = np.asarray(normals, np.float32)
    uvs = np.asarray(uvs, np.float32).copy()
    material_indices = np.asarray(material_indices, np.uint8)
    T = positions.shape[0]
    flat = positions.reshape(-1, 3).astype(np.float64)
    box_min, box_max = flat.min(axis=0), flat.max(axis=0)
    box_max = np.maximum(box_max, box_min + 1e-6)
    if sort_triangles:
        c = positions.astype(np.float64).mean(axis=1)
        g = np.clip(((c - c.min(axis=0)) / np.maximum(np.ptp(c, axis=0), 1e-12) * 1023.0), 0, 1023).astype(np.uint64)
        code = (_morton_10(g[:, 0]) << np.uint64(2)) | (_morton_10(g[:, 1]) << np.uint64(1)) | _morton_10(g[:, 2])
        order = np.argsort(code, kind="stable")
        positions, normals, uvs, material_indices = positions[order], normals[order], uvs[order], material_indices[order]
        flat = positions.reshape(-1, 3).astype(np.float64)
    if shuffle_seed is not None:
        order = np.random.default_rng(shuffle_seed).permutation(T)
        positions, normals, uvs, material_indices = positions[order], normals[order], uvs[order], material_indices[order]
        flat = positions.reshape(-1, 3).astype(np.float64)
    # 21 bits per coordinate; the summand places samples at cell centres
    qf = (2.0 ** 21) / (box_max - box_min)
    q = np.minimum((flat * qf - box_min * qf).astype(np.uint64), 2 ** 21 - 1).astype(np.uint32)
    factor = (1.0 / qf).astype(np.float32)
    summand = (box_min + 0.5 / qf).astype(np.float32)
    packed = np.zeros((T * 3, 2), np.uint32)
    packed[:, 0] = q[:, 0] | ((q[:, 1] & 0x7FF) << 21)
    packed[:, 1] = ((q[:, 1] & 0x1FF800) >> 11) | (q[:, 2] << 10)
    # texture coordinates: shift each triangle into [0, 8), 16-bit UNORM of uv / 8
    uvs -= np.floor(uvs.min(axis=1, keepdims=True))
    nuv = np.zeros((T * 3, 4), np.uint16)
    nuv[:, 2:4] = np.clip(uvs.reshape(-1, 2) * (65535.0 / 8.0) + 0.5, 0.0, 65535.0).astype(np.uint16)
    nx, ny = encode_octahedral_normals(normals.reshape(-1, 3))
    nuv[:, 0], nuv[:, 1] = nx, ny
    with open(path, "wb") as f:
        f.write(struct.pack("<II", 0x00ABCABC, 1))
        f.write(struct.pack("<QQ", len(material_names), T))
        f.write(struct.pack("<fff", *factor))
        f.write(struct.pack("<fff", *summand))
        for name in material_names:
            b = name.encode("utf-8")
            f.write(struct.pack("<Q", len(b)))
            f.write(b + b"\0")
        f.write(packed.tobytes())
        f.write(nuv.tobytes())
        f.write(material_indices.tobytes())
        f.write(struct.pack("<I", 0x00E0FE0F))
    return {"quantized_positions": packed, "normals_and_tex_coords": nuv, "material_indices": material_indices,
            "dequantization_factor": factor, "dequantization_summand": summand}


def _box_triangles(center, half, rotation_z):
    c, s = math.cos(rotation_z), math.sin(rotation_z)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    corners = np.array([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], np.float64) * half
    corners = corners @ R.T + center
    # faces with outward counter-clockwise winding (corner index = x + 2y + 4z)
    faces = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]
    tris, nrms = [], []
    for f in faces:
        p = corners[list(f)]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        n /= np.linalg.norm(n)
        for a, b, c2 in ((0, 1, 2), (0, 2, 3)):
            tris.append([p[a], p[b], p[c2]])
            nrms.append([n, n, n])
    return np.array(tris), np.array(nrms)


def make_scene_geometry(grid=256, box_count=64, seed=1234, extent=10.0, materials=3):
    """Ground plane [-extent, extent]^2 at z = 0 with 2 * grid^2 triangles plus
    `box_count` random boxes standing on it.  Returns positions, normals, uvs,
    material indices."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(-extent, extent, grid + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    p00 = np.stack([X[:-1, :-1], Y[:-1, :-1], np.zeros((grid, grid))], -1)
    p10 = np.stack([X[:-1, 1:], Y[:-1, 1:], np.zeros((grid, grid))], -1)
    p01 = np.stack([X[1:, :-1], Y[1:, :-1], np.zeros((grid, grid))], -1)
    p11 = np.stack([X[1:, 1:], Y[1:, 1:], np.zeros((grid, grid))], -1)
    t0 = np.stack([p00, p10, p11], -2).reshape(-1, 3, 3)
    t1 = np.stack([p00, p11, p01], -2).reshape(-1, 3, 3)
    positions = [np.concatenate([t0, t1], 0)]
    normals = [np.tile(np.array([0.0, 0.0, 1.0]), (2 * grid * grid, 3, 1))]
    # material stripes on the ground echo the reference's "roughness planes" scene
    stripe = ((positions[0][:, :, 1].mean(axis=1) + extent) / (2 * extent) * materials).astype(np.int64)
    mats = [np.clip(stripe, 0, materials - 1)]
    for b in range(box_count):
        half = rng.uniform(0.15, 0.6, 3)
        center = np.array([rng.uniform(-extent * 0.8, extent * 0.8), rng.uniform(-extent * 0.8, extent * 0.8), half[2]])
        t, n = _box_triangles(center, half, rng.uniform(0, math.pi))
        positions.append(t)
        normals.append(n)
        mats.append(np.full(len(t), b % materials))
    positions = np.concatenate(positions, 0)
    normals = np.concatenate(normals, 0)
    mats = np.concatenate(mats, 0)
    uvs = positions[:, :, :2] * 0.5
    return positions, normals, uvs, mats


def _boxes_triangles(centers, halves, rotations):
    """Vectorised _box_triangles: (n, 3) centres and half extents, (n,) rotations about z -> (12 n, 3, 3) positions and normals."""
    centers, halves, rotations = np.asarray(centers, np.float64), np.asarray(halves, np.float64), np.asarray(rotations, np.float64)
    n = len(centers)
    c, s = np.cos(rotations), np.sin(rotations)
    R = np.zeros((n, 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s, s, c, 1.0
    unit = np.array([[x, y, z] for z in (-1, 1) for y in (-1, 1) for x in (-1, 1)], np.float64)
    corners = np.einsum("nij,nkj->nki", R, unit[None] * halves[:, None, :]) + centers[:, None, :]
    faces = np.array([(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)])
    quads = corners[:, faces]                                   # (n, 6, 4, 3)
    normal = np.cross(quads[:, :, 1] - quads[:, :, 0], quads[:, :, 2] - quads[:, :, 0])
    normal /= np.linalg.norm(normal, axis=-1, keepdims=True)
    tris = np.stack([quads[:, :, [0, 1, 2]], quads[:, :, [0, 2, 3]]], 2).reshape(-1, 3, 3)
    normals = np.repeat(np.repeat(normal[:, :, None, None, :], 2, 2), 3, 3).reshape(-1, 3, 3)
    return tris, normals


def _sphere_triangles(center, radius, segments, rings):
    """UV sphere: 2 * segments * (rings - 1) triangles (thin ones at the poles) with smooth normals."""
    theta = np.linspace(0.0, math.pi, rings + 1)
    phi = np.linspace(0.0, 2.0 * math.pi, segments + 1)
    T, P = np.meshgrid(theta, phi, indexing="ij")
    unit = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1)
    a, b, c, d = unit[:-1, :-1], unit[1:, :-1], unit[1:, 1:], unit[:-1, 1:]
    upper = np.stack([a, b, c], -2)[:-1].reshape(-1, 3, 3)     # (b = c at the last ring, a = d at the first: those would be degenerate)
    lower = np.stack([a, c, d], -2)[1:].reshape(-1, 3, 3)
    normals = np.concatenate([upper, lower], 0)
    return normals * radius + np.asarray(center, np.float64), normals


def make_large_scene_geometry(grid=768, tower_count=700, sphere_count=160, fence_count=220, louvre_count=36, seed=4321, extent=10.0, materials=8):
    """The second, harder scene ("any scene but one"): 2 - 3 M triangles with what the benchmark
    scene lacks - stacked occluders (towers of boxes), long thin triangles (fence bars 4 cm x 2.4 m, louvre slats
    3 m x 6 cm hanging between the floor and the lights), small dense meshes next to large sparse ones (spheres of
    8 k triangles on a floor of 2 * grid^2), deep occlusion (most shadow rays pass several fences) and several
    materials.  Same lights and camera as the benchmark scene.  Returns positions, normals, uvs, material indices."""
    rng = np.random.default_rng(seed)
    positions, normals, uvs, mats = make_scene_geometry(grid, 0, seed, extent, materials)
    positions, normals, mats = [positions], [normals], [mats]
    camera_xy = np.array(DEFAULT_CAMERA["position"][:2])

    def place(low, high):
        """a position in [low, high]^2 that leaves 1.5 m around the camera free"""
        while True:
            xy = rng.uniform(low, high, 2)
            if np.linalg.norm(xy - camera_xy) > 1.5:
                return xy

    def add(tris, nrms, material):
        positions.append(tris)
        normals.append(nrms)
        mats.append(np.full(len(tris), material) if np.isscalar(material) else np.asarray(material))

    # towers: three to eight boxes on top of each other, shrinking and turning
    for _ in range(tower_count):
        x, y = place(-extent * 0.9, extent * 0.9)
        levels = int(rng.integers(3, 9))
        half = rng.uniform(0.12, 0.45, 3) * np.array([1.0, 1.0, 0.5])
        z, centers, halves, rotations = 0.0, [], [], []
        rotation = rng.uniform(0, math.pi)
        for _level in range(levels):
            centers.append((x, y, z + half[2]))
            halves.append(half.copy())
            rotations.append(rotation)
            z += 2.0 * half[2]
            half = half * rng.uniform(0.7, 0.95)
            rotation += rng.uniform(0.1, 0.6)
        tris, nrms = _boxes_triangles(centers, halves, rotations)
        add(tris, nrms, int(rng.integers(0, materials)))
    # fences: rows of thin vertical bars (each bar a box 4 cm x 4 cm x up to 2.4 m: twelve long thin triangles)
    for _ in range(fence_count):
        x, y = place(-extent * 0.85, extent * 0.85)
        angle, bars, height = rng.uniform(0, math.pi), int(rng.integers(12, 40)), rng.uniform(1.2, 2.4)
        step = np.array([math.cos(angle), math.sin(angle)]) * 0.11
        offsets = (np.arange(bars) - 0.5 * bars)[:, None] * step
        centers = np.concatenate([np.array([x, y]) + offsets, np.full((bars, 1), 0.5 * height)], 1)
        halves = np.tile(np.array([0.02, 0.02, 0.5 * height]), (bars, 1))
        tris, nrms = _boxes_triangles(centers, halves, np.full(bars, angle))
        add(tris, nrms, int(rng.integers(0, materials)))
    # louvres: horizontal slats (3 m x 6 cm x 1 cm, tilted) in layers between the floor and the lights
    for _ in range(louvre_count):
        x, y = rng.uniform(-4.5, 4.5), rng.uniform(-1.0, 6.5)
        angle, slats, z = rng.uniform(0, math.pi), int(rng.integers(10, 28)), rng.uniform(1.1, 2.0)
        step = np.array([-math.sin(angle), math.cos(angle)]) * 0.13
        offsets = (np.arange(slats) - 0.5 * slats)[:, None] * step
        centers = np.concatenate([np.array([x, y]) + offsets, np.full((slats, 1), z)], 1)
        halves = np.tile(np.array([rng.uniform(0.8, 1.5), 0.03, 0.005]), (slats, 1))
        tris, nrms = _boxes_triangles(centers, halves, np.full(slats, angle))
        add(tris, nrms, int(rng.integers(0, materials)))
    # spheres: 64 x 64 segments, i.e. 8 064 small triangles each
    for _ in range(sphere_count):
        radius = rng.uniform(0.12, 0.5)
        center = (*place(-extent * 0.85, extent * 0.85), radius * rng.uniform(1.0, 2.5))
        tris, nrms = _sphere_triangles(center, radius, 64, 64)
        add(tris, nrms, int(rng.integers(0, materials)))
    positions = np.concatenate(positions, 0)
    normals = np.concatenate(normals, 0)
    mats = np.concatenate(mats, 0)
    uvs = positions[:, :, :2] * 0.5
    return positions, normals, uvs, mats


# ---- .vkt constant textures --------------------------------------------------------

def write_constant_vkt(path, rgba):
    """1x1 RGBA32F texture (VkFormat 109)."""
    payload = struct.pack("<ffff", *rgba)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiiii", 0xBC1BC1, 1, 1, 1, 1, 109))
        f.write(struct.pack("<Q", len(payload)))
        f.write(struct.pack("<iiQQ", 1, 1, len(payload), 0))
        f.write(payload)
        f.write(struct.pack("<I", 0xE0FE0F))


DEFAULT_MATERIALS = {
    # name: (base colour, (occlusion, linear roughness, metalicity))
    "rough_grey": ((0.8, 0.8, 0.8), (1.0, 0.7, 0.0)),
    "glossy_red": ((0.7, 0.25, 0.2), (1.0, 0.35, 0.0)),
    "brushed_metal": ((0.9, 0.8, 0.6), (1.0, 0.25, 1.0)),
}


LARGE_SCENE_MATERIALS = dict(DEFAULT_MATERIALS, **{
    "matte_blue": ((0.2, 0.3, 0.7), (1.0, 0.8, 0.0)),
    "polished_green": ((0.2, 0.6, 0.3), (1.0, 0.15, 0.0)),
    "copper": ((0.95, 0.64, 0.54), (1.0, 0.3, 1.0)),
    "chalk": ((0.9, 0.9, 0.85), (1.0, 0.95, 0.0)),
    "dark_steel": ((0.4, 0.42, 0.45), (1.0, 0.4, 1.0)),
})


def write_material_textures(directory, materials=None):
    materials = materials or DEFAULT_MATERIALS
    os.makedirs(directory, exist_ok=True)
    for name, (base, spec) in materials.items():
        write_constant_vkt(os.path.join(directory, name + "_BaseColor.vkt"), (*base, 1.0))
        write_constant_vkt(os.path.join(directory, name + "_Specular.vkt"), (*spec, 1.0))
        write_constant_vkt(os.path.join(directory, name + "_Normal.vkt"), (0.5, 0.5, 1.0, 1.0))
    return list(materials.keys())


# ---- textured materials: *.vkt with 8-bit and block-compressed mip chains ------------------

VK_FORMAT_R8G8B8A8_UNORM, VK_FORMAT_R8G8B8A8_SRGB = 37, 43
VK_FORMAT_BC1_RGB_UNORM, VK_FORMAT_BC1_RGB_SRGB, VK_FORMAT_BC5_UNORM = 131, 132, 141


def write_vkt(path, vk_format, extents, payloads):
    """The container of reference src/textures.c:95-129: header, mip table, data, end marker."""
    offsets, offset = [], 0
    for data in payloads:
        offsets.append(offset)
        offset += len(data)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiiii", 0xBC1BC1, 1, len(payloads), extents[0][0], extents[0][1], vk_format))
        f.write(struct.pack("<Q", offset))
        for (w, h), data, o in zip(extents, payloads, offsets):
            f.write(struct.pack("<iiQQ", w, h, len(data), o))
        for data in payloads:
            f.write(data)
        f.write(struct.pack("<I", 0xE0FE0F))


def mip_chain(image):
    """Box-filtered chain down to 1x1 (uint8, any channel count; extents halve and floor)."""
    chain = [np.ascontiguousarray(image, np.uint8)]
    while chain[-1].shape[0] > 1 or chain[-1].shape[1] > 1:
        a = chain[-1].astype(np.float64)
        h, w = max(a.shape[0] // 2, 1), max(a.shape[1] // 2, 1)
        a = a[:2 * h if a.shape[0] > 1 else 1, :2 * w if a.shape[1] > 1 else 1]
        if a.shape[0] > 1:
            a = 0.5 * (a[0::2] + a[1::2])
        if a.shape[1] > 1:
            a = 0.5 * (a[:, 0::2] + a[:, 1::2])
        chain.append(np.clip(np.rint(a), 0, 255).astype(np.uint8))
    return chain


def _pad_to_blocks(image):
    h, w = image.shape[:2]
    return np.pad(image, ((0, (-h) % 4), (0, (-w) % 4), (0, 0)), mode="edge")


def _to_565(rgb):
    return ((int(rgb[0]) >> 3) << 11) | ((int(rgb[1]) >> 2) << 5) | (int(rgb[2]) >> 3)


def decode_bc1_block(block, has_alpha=False):
    """Reference decoder for the tests: the format definition with endpoints expanded by bit
    replication and interpolants rounded to nearest (the choice csrc/host/textures.c documents)."""
    c0, c1, indices = struct.unpack("<HHI", block)
    def expand(c):
        r, g, b = c >> 11, (c >> 5) & 63, c & 31
        return [(r << 3) | (r >> 2), (g << 2) | (g >> 4), (b << 3) | (b >> 2), 255]
    colors = [expand(c0), expand(c1), None, None]
    if c0 > c1:
        colors[2] = [(2 * a + b + 1) // 3 for a, b in zip(colors[0][:3], colors[1][:3])] + [255]
        colors[3] = [(a + 2 * b + 1) // 3 for a, b in zip(colors[0][:3], colors[1][:3])] + [255]
    else:
        colors[2] = [(a + b + 1) // 2 for a, b in zip(colors[0][:3], colors[1][:3])] + [255]
        colors[3] = [0, 0, 0, 0 if has_alpha else 255]
    return np.array([colors[(indices >> (2 * i)) & 3] for i in range(16)], np.uint8).reshape(4, 4, 4)


def decode_bc4_block(block):
    e0, e1 = block[0], block[1]
    values = [e0, e1]
    if e0 > e1:
        values += [((7 - i) * e0 + i * e1 + 3) // 7 for i in range(1, 7)]
    else:
        values += [((5 - i) * e0 + i * e1 + 2) // 5 for i in range(1, 5)] + [0, 255]
    indices = int.from_bytes(block[2:8], "little")
    return np.array([values[(indices >> (3 * i)) & 7] for i in range(16)], np.uint8).reshape(4, 4)


def encode_bc1(image):
    """A simple BC1 encoder (endpoints = extreme colours of the block by luminance)."""
    padded = _pad_to_blocks(image[..., :3])
    out = bytearray()
    for by in range(0, padded.shape[0], 4):
        for bx in range(0, padded.shape[1], 4):
            block = padded[by:by + 4, bx:bx + 4].reshape(16, 3).astype(np.int32)
            luminance = block @ np.array([2, 5, 1])
            c0, c1 = _to_565(block[luminance.argmax()]), _to_565(block[luminance.argmin()])
            if c0 < c1:
                c0, c1 = c1, c0
            if c0 == c1:
                out += struct.pack("<HHI", c0, c1, 0)
                continue
            palette = decode_bc1_block(struct.pack("<HHI", c0, c1, 0x1B))  # texels 0..3 use indices 3, 2, 1, 0
            palette = {3: palette[0, 0, :3], 2: palette[0, 1, :3], 1: palette[0, 2, :3], 0: palette[0, 3, :3]}
            indices = 0
            for i, texel in enumerate(block):
                best = min(range(4), key=lambda k: int(((palette[k].astype(np.int32) - texel) ** 2).sum()))
                indices |= best << (2 * i)
            out += struct.pack("<HHI", c0, c1, indices)
    return bytes(out)


def encode_bc4(channel):
    padded = _pad_to_blocks(channel[..., None])[..., 0]
    out = bytearray()
    for by in range(0, padded.shape[0], 4):
        for bx in range(0, padded.shape[1], 4):
            block = padded[by:by + 4, bx:bx + 4].reshape(16).astype(np.int32)
            e0, e1 = int(block.max()), int(block.min())
            if e0 == e1:
                out += bytes([e0, e1]) + bytes(6)
                continue
            values = decode_bc4_block(bytes([e0, e1]) + (0o76543210).to_bytes(3, "little") * 2)[0].astype(np.int32)  # texels 0..7 use indices 0..7
            indices = 0
            for i, texel in enumerate(block):
                indices |= int(np.abs(values[:8] - texel).argmin()) << (3 * i)
            out += bytes([e0, e1]) + indices.to_bytes(6, "little")
    return bytes(out)


def encode_bc5(image):
    red, green = encode_bc4(image[..., 0]), encode_bc4(image[..., 1])
    return b"".join(red[i:i + 8] + green[i:i + 8] for i in range(0, len(red), 8))


def procedural_textures(size=64, seed=3):
    """Base colour (sRGB), specular (occlusion, linear roughness, metalicity) and tangent-space
    normal (xy in [0, 1]) images with features at several scales, so that every mip level matters."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    checker = ((np.floor(x * 8) + np.floor(y * 8)) % 2)
    noise = rng.random((size, size))
    base = np.stack([0.25 + 0.6 * checker, 0.3 + 0.5 * x, 0.2 + 0.6 * noise], -1)
    specular = np.stack([np.ones_like(x), 0.25 + 0.5 * (0.5 + 0.5 * np.sin(12.0 * x) * np.cos(9.0 * y)), 0.3 * checker], -1)
    normal = np.stack([0.5 + 0.12 * np.sin(20.0 * x), 0.5 + 0.12 * np.cos(16.0 * y), np.ones_like(x)], -1)
    to8 = lambda a: np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8)
    return to8(base), to8(specular), to8(normal)


def write_textured_material_textures(directory, names, size=64, formats=("bc1_srgb", "rgba8", "bc5")):
    """*.vkt files with real images for every material: BC1 sRGB base colour, RGBA8 specular,
    BC5 normal by default (what the reference's converter produces); mip chains down to 1x1."""
    os.makedirs(directory, exist_ok=True)
    for index, name in enumerate(names):
        base, specular, normal = procedural_textures(size, seed=3 + index)
        for image, suffix, kind in ((base, "BaseColor", formats[0]), (specular, "Specular", formats[1]), (normal, "Normal", formats[2])):
            rgba = np.concatenate([image, np.full(image.shape[:2] + (1,), 255, np.uint8)], -1)
            chain = mip_chain(rgba)
            extents = [(m.shape[1], m.shape[0]) for m in chain]
            if kind.startswith("bc1"):
                payloads, vk_format = [encode_bc1(m) for m in chain], (VK_FORMAT_BC1_RGB_SRGB if kind.endswith("srgb") else VK_FORMAT_BC1_RGB_UNORM)
            elif kind == "bc5":
                payloads, vk_format = [encode_bc5(m) for m in chain], VK_FORMAT_BC5_UNORM
            else:
                payloads, vk_format = [m.tobytes() for m in chain], (VK_FORMAT_R8G8B8A8_SRGB if kind.endswith("srgb") else VK_FORMAT_R8G8B8A8_UNORM)
            write_vkt(os.path.join(directory, "%s_%s.vkt" % (name, suffix)), vk_format, extents, payloads)


# ---- light textures: half / float *.vkt as the reference's converter writes for probes and IES ----

VK_FORMAT_R16G16B16_SFLOAT, VK_FORMAT_R16G16B16A16_SFLOAT = 90, 97
VK_FORMAT_R32G32B32_SFLOAT, VK_FORMAT_R32G32B32A32_SFLOAT = 106, 109


def light_texture_image(kind, width=64, height=32, seed=11):
    """float64 RGB image (height, width, 3): "area" a tiled emblem, "portal" a sky-like probe with a
    hot spot (values above 1), "ies" a one-row intensity profile."""
    rng = np.random.default_rng(seed)
    if kind == "ies":
        height = 1
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    u, v = (x + 0.5) / width, (y + 0.5) / height
    if kind == "area":
        checker = (np.floor(u * 6) + np.floor(v * 4)) % 2
        return np.stack([0.2 + 0.8 * checker, 0.3 + 0.7 * u, 0.4 + 0.6 * (1.0 - v)], -1) * (0.6 + 0.4 * rng.random((height, width, 1)))
    if kind == "portal":
        sun = 6.0 * np.exp(-((u - 0.3) ** 2 + (v - 0.35) ** 2) * 90.0)
        sky = np.stack([0.3 + 0.2 * v, 0.4 + 0.3 * v, 0.9 - 0.4 * v], -1)
        return sky * (0.8 + 0.2 * np.sin(14.0 * np.pi * u)[..., None]) + sun[..., None]
    profile = (0.15 + np.cos(0.5 * np.pi * u) ** 2 * (1.0 + 0.5 * np.sin(9.0 * np.pi * u))) * 1.5
    return np.stack([profile, profile, profile], -1)


def write_light_texture(path, image, vk_format=VK_FORMAT_R16G16B16A16_SFLOAT, mips=2):
    """Writes image (float, (h, w, 3)) as a half / float *.vkt with `mips` box-filtered levels."""
    levels = [np.asarray(image, np.float64)]
    for _ in range(mips - 1):
        a = levels[-1]
        if a.shape[0] > 1:
            a = 0.5 * (a[0:2 * (a.shape[0] // 2):2] + a[1::2])
        if a.shape[1] > 1:
            a = 0.5 * (a[:, 0:2 * (a.shape[1] // 2):2] + a[:, 1::2])
        levels.append(a)
    dtype = np.float16 if vk_format in (VK_FORMAT_R16G16B16_SFLOAT, VK_FORMAT_R16G16B16A16_SFLOAT) else np.float32
    payloads = []
    for a in levels:
        if vk_format in (VK_FORMAT_R16G16B16A16_SFLOAT, VK_FORMAT_R32G32B32A32_SFLOAT):
            a = np.concatenate([a, np.ones(a.shape[:2] + (1,))], -1)
        payloads.append(np.ascontiguousarray(a, dtype).tobytes())
    write_vkt(path, vk_format, [(a.shape[1], a.shape[0]) for a in levels], payloads)


# ---- LTC fit files ------------------------------------------------------------------

def write_ltc_fits(directory, resolution=32, fresnel_count=51):
    """Writes fit<i>.dat files with a smooth synthetic GGX-like fit.  NOT a real
    LTC fit (those are downloads, reference README.md:9-13): the matrix is
    diag-dominant with a lobe that narrows with roughness and tilts away from the
    viewer at grazing angles, which is all the parity tests need.  File layout:
    u64 resolution, then resolution^2 x (d0, d1, d2, d3, albedo) float32 with
    index y * R + x, x = roughness axis, y = inclination axis."""
    os.makedirs(directory, exist_ok=True)
    R = resolution
    x = (np.arange(R) / (R - 1)) ** 2          # roughness alpha (table axis is sqrt(alpha))
    theta = np.arange(R) / (R - 1) * (0.5 * math.pi)
    A, TH = np.meshgrid(np.maximum(x, 0.0064), theta, indexing="xy")
    sin_t, cos_t = np.sin(TH), np.cos(TH)
    # cosine -> shading matrix [[a, 0, b], [0, c, 0], [e, 0, 1]]; the file stores
    # (d0, d1, d2, d3) = (a, e, c, b) (reference ltc_table.c:86-90 + ltc_utility.glsl:71-74)
    a = A * (1.0 + 0.8 * sin_t ** 2) + 0.02
    c = A * (1.0 + 0.2 * sin_t ** 2) + 0.02
    b = -(1.0 - A) ** 2 * sin_t / np.maximum(cos_t, 0.25) * 0.6
    e = 0.15 * A * sin_t
    for i in range(fresnel_count):
        f0 = i / max(fresnel_count - 1, 1)
        fres = f0 + (1.0 - f0) * (1.0 - cos_t) ** 5
        albedo = np.clip(fres * (1.0 - 0.35 * A), 0.0, 1.0)
        data = np.stack([a, e, c, b, albedo], -1).astype(np.float32)
        with open(os.path.join(directory, "fit%d.dat" % i), "wb") as f:
            f.write(struct.pack("<Q", R))
            f.write(data.tobytes())


# ---- lights and cameras of the BASELINE configurations ---------------------------------

def regular_polygon(n, radius=0.5, phase=0.0):
    ang = phase + np.arange(n) * (2.0 * math.pi / n)
    return np.stack([0.5 + radius * np.cos(ang), 0.5 + radius * np.sin(ang)], -1).astype(np.float32)


def light_spec(vertices_plane, translation, rotation_angles, flux=(10.0, 10.0, 10.0), scaling=(1.0, 1.0)):
    return {"vertices_plane_space": np.asarray(vertices_plane, np.float32), "translation": tuple(translation),
            "rotation_angles": tuple(rotation_angles), "radiant_flux": tuple(flux), "scaling": tuple(scaling)}


DEFAULT_CAMERA = {"position": (-3.0, -2.0, 1.65), "rotation_x": 0.43 * math.pi, "rotation_z": 1.3 * math.pi,
                  "vertical_fov": 0.33 * math.pi, "near": 0.05, "far": 1.0e3}

QUAD = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)]


def config_lights(config):
    """Light set-ups for BASELINE.json configs 1-4."""
    pi = math.pi
    if config == 1:
        return [light_spec([(0, 0), (1, 0), (0, 1)], (-1.0, 0.5, 3.0), (pi, 0.0, 0.0), (10, 10, 10), (1.5, 1.5))]
    if config == 2:
        return [light_spec(regular_polygon(5), (-0.5, 1.0, 2.5), (0.85 * pi, 0.1, 0.3), (12, 11, 10), (1.6, 1.6))]
    if config == "target":
        # north_star's target shape: one light of config 3
        return config_lights(3)[:1]
    if config == 3:
        out = []
        for k, (tx, ty) in enumerate([(-1.5, 1.5), (1.5, 1.5), (-1.5, 4.0), (1.5, 4.0)]):
            out.append(light_spec(QUAD, (tx, ty, 2.2 + 0.2 * k), (0.5 * pi + 0.45 * (k % 2) + 0.3, 0.0, 0.4 * k),
                                  (6 + k, 6, 8 - k), (1.0, 0.8)))
        return out
    if config == 4:
        out = []
        rng = np.random.default_rng(77)
        for k in range(8):
            n = 3 + (k % 4)
            out.append(light_spec(regular_polygon(n, 0.5, 0.2 * k),
                                  (rng.uniform(-4, 4), rng.uniform(0, 7), rng.uniform(1.5, 3.5)),
                                  (rng.uniform(0.6, 1.0) * pi, rng.uniform(-0.3, 0.3), rng.uniform(0, 2 * pi)),
                                  tuple(rng.uniform(3, 9, 3)), (rng.uniform(0.7, 1.5), rng.uniform(0.7, 1.5))))
        return out
    raise ValueError("unknown config %r" % (config,))


CONFIG_SETTINGS = {
    # BASELINE.json configs; S = samples per technique per light
    1: dict(width=512, height=512, sample_count=1, sampling_strategies="diffuse_only", mis_heuristic="balance",
            polygon_technique="projected_solid_angle", trace_shadow_rays=False),
    2: dict(width=1920, height=1080, sample_count=1, sampling_strategies="diffuse_ggx_mis", mis_heuristic="balance",
            polygon_technique="projected_solid_angle", trace_shadow_rays=True),
    3: dict(width=1920, height=1080, sample_count=4, sampling_strategies="diffuse_specular_mis", mis_heuristic="optimal_clamped",
            polygon_technique="projected_solid_angle", trace_shadow_rays=True),
    4: dict(width=3840, height=2160, sample_count=8, sampling_strategies="diffuse_specular_mis", mis_heuristic="optimal_clamped",
            polygon_technique="projected_solid_angle", trace_shadow_rays=True),
    # BASELINE.json north_star target: ">= 1 Gsample/s at 1920x1080, 4 spp, 1 polygonal light"
    "target": dict(width=1920, height=1080, sample_count=4, sampling_strategies="diffuse_specular_mis", mis_heuristic="optimal_clamped",
                   polygon_technique="projected_solid_angle", trace_shadow_rays=True),
}


def write_dataset(directory, grid=256, box_count=64, seed=1234, ltc_resolution=32, fresnel_count=51, textured=False, texture_size=64, shuffle_seed=None, large=None):
    """Writes scene.vks, textures/, ltc/ below `directory` and returns the paths.  textured:
    real images (BC1 / RGBA8 / BC5 with mip chains) instead of constant material textures.
    large: a dict of make_large_scene_geometry() arguments ({} for its defaults: 2.7 M triangles, eight
    materials) instead of the benchmark scene of make_scene_geometry(grid, box_count)."""
    os.makedirs(directory, exist_ok=True)
    names = write_material_textures(os.path.join(directory, "textures"), LARGE_SCENE_MATERIALS if large is not None else None)
    if textured:
        write_textured_material_textures(os.path.join(directory, "textures"), names, texture_size)
    if large is not None:
        positions, normals, uvs, mats = make_large_scene_geometry(**dict({"seed": seed, "materials": len(names)}, **large))
    else:
        positions, normals, uvs, mats = make_scene_geometry(grid, box_count, seed, materials=len(names))
    scene_path = os.path.join(directory, "scene.vks")
    write_vks(scene_path, positions, normals, uvs, mats, names, shuffle_seed=shuffle_seed)
    ltc_dir = os.path.join(directory, "ltc")
    write_ltc_fits(ltc_dir, ltc_resolution, fresnel_count)
    return {"scene": scene_path, "textures": os.path.join(directory, "textures"), "ltc": ltc_dir, "fresnel_count": fresnel_count,
            "light_textures": write_light_textures(os.path.join(directory, "light_textures"))}


def write_light_textures(directory):
    """One texture per texturing technique of the reference (polygonal_light.h:75-90), in the
    formats its converter produces: an sRGB BC1 emblem with a mip chain, a half-float probe,
    a one-row fp32 IES profile, and the probe once more as three-channel half floats."""
    os.makedirs(directory, exist_ok=True)
    paths = {name: os.path.join(directory, name + ".vkt") for name in ("area", "portal", "ies", "portal_rgb16")}
    emblem = np.clip(np.rint(light_texture_image("area", 32, 32) * 255.0), 0, 255).astype(np.uint8)
    chain = mip_chain(np.concatenate([emblem, np.full(emblem.shape[:2] + (1,), 255, np.uint8)], -1))
    write_vkt(paths["area"], VK_FORMAT_BC1_RGB_SRGB, [(m.shape[1], m.shape[0]) for m in chain], [encode_bc1(m) for m in chain])
    write_light_texture(paths["portal"], light_texture_image("portal", 64, 32), VK_FORMAT_R16G16B16A16_SFLOAT, mips=3)
    write_light_texture(paths["ies"], light_texture_image("ies", 48, 1), VK_FORMAT_R32G32B32_SFLOAT, mips=1)
    write_light_texture(paths["portal_rgb16"], light_texture_image("portal", 40, 20, seed=5), VK_FORMAT_R16G16B16_SFLOAT, mips=2)
    return paths
