#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Compiles the reference's shading-pass GLSL as C++.

The shader sources are read where they lie (--reference/src/shaders), run through
a handful of purely syntactic substitutions (listed in PATCHES below) into a
temporary directory and compiled with g++ against glsl_compat.hpp and
ref_shader_main.cpp.  One shared object is produced per shader variant, with the
same preprocessor defines the reference hands to glslangValidator
(src/main.c:752-792).  Nothing derived from the reference is written into the
repository; outputs go to oracle/_ref (git-ignored).
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))

SHADER_FILES = ["shading_pass.frag.glsl", "polygon_sampling.glsl", "polygon_clipping.glsl", "ltc_utility.glsl",
                "brdfs.glsl", "noise_utility.glsl", "mesh_quantization.glsl", "polygonal_light_utility.glsl",
                "shared_constants.glsl", "srgb_utility.glsl", "math_constants.glsl", "unrolling.glsl",
                "polygon_sampling_related_work.glsl", "cubic_solver.glsl"]


def translate(name, text):
    """GLSL -> C++ at the token level.  Every rule is syntactic; no arithmetic is touched."""
    out = []
    in_uniform_block = False
    if name == "polygon_sampling_related_work.glsl":
        pass  # the whole related-work file is in scope
    for line in text.split("\n"):
        s = line.strip()
        # directives that mean nothing to a C++ compiler
        if s.startswith("#version") or s.startswith("#extension"):
            continue
        # uniform block -> plain globals
        if re.match(r"layout\s*\(std140.*\)\s*uniform\s+per_frame_constants\s*\{", s):
            in_uniform_block = True
            continue
        if in_uniform_block and s == "};":
            in_uniform_block = False
            continue
        # single-line resource declarations: drop the layout qualifier and storage class
        m = re.match(r"\s*layout\s*\([^)]*\)\s*(uniform|in|out)\s+(.*)$", line)
        if m:
            decl = m.group(2)
            if decl.startswith("vec4 gl_FragCoord") or decl.startswith("vec4 g_out_color"):
                decl = "thread_local " + decl
            line = decl
        # control-flow attributes
        line = re.sub(r"\[\[(unroll|dont_unroll)\]\]", "", line)
        # parameter qualifiers: arrays already decay to pointers, everything else becomes a reference
        line = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)\s*\[", r"\1 \2[", line)
        line = re.sub(r"\b(?:inout|out)\s+(\w+)\s+(\w+)", r"\1& \2", line)
        out.append(line)
    text = "\n".join(out)
    if name == "shading_pass.frag.glsl":
        text = text.replace("void main()", "void shader_main()")
        # GLSL converts between integer vector types implicitly; spell the conversions out
        text = text.replace("ivec2(gl_FragCoord.xy)", "to_ivec2(gl_FragCoord.xy)")
        text = text.replace("vec3(pixel, 1.0f)", "make_vec3(pixel, 1.0f)")
        text = text.replace("get_noise_accessor(pixel,", "get_noise_accessor(uvec2((uint) pixel.x, (uint) pixel.y),")
        # the clamped UBO array size may be zero-length in C++ terms only if no light exists
    if name == "noise_utility.glsl":
        text = text.replace("ivec3(sample_location, texture_index)", "make_ivec3(sample_location, texture_index)")
        # C++ cannot unify two different swizzle proxy types in ?:
        text = text.replace("? random_numbers.yzw : random_numbers.xyz", "? uvec3(random_numbers.yzw) : uvec3(random_numbers.xyz)")
    return text


def variant_defines(v):
    strategies = ["DIFFUSE_ONLY", "DIFFUSE_GGX_MIS", "DIFFUSE_SPECULAR_SEPARATELY", "DIFFUSE_SPECULAR_MIS", "DIFFUSE_SPECULAR_RANDOM"]
    heuristics = ["BALANCE", "POWER", "WEIGHTED", "OPTIMAL_CLAMPED", "OPTIMAL"]
    techniques = {"baseline": "BASELINE", "area_turk": "AREA_TURK", "solid_angle_arvo": "SOLID_ANGLE_ARVO",
                  "rectangle_solid_angle_urena": "RECTANGLE_SOLID_ANGLE_URENA", "solid_angle": "SOLID_ANGLE",
                  "clipped_solid_angle": "CLIPPED_SOLID_ANGLE", "bilinear_cosine_warp_hart": "BILINEAR_COSINE_WARP_HART",
                  "bilinear_cosine_warp_clipping_hart": "BILINEAR_COSINE_WARP_CLIPPING_HART",
                  "biquadratic_cosine_warp_hart": "BIQUADRATIC_COSINE_WARP_HART",
                  "biquadratic_cosine_warp_clipping_hart": "BIQUADRATIC_COSINE_WARP_CLIPPING_HART",
                  "projected_solid_angle_arvo": "PROJECTED_SOLID_ANGLE_ARVO", "projected_solid_angle": "PROJECTED_SOLID_ANGLE"}
    technique = v.get("technique", "projected_solid_angle")
    biased = technique == "projected_solid_angle_biased"
    if biased:
        technique = "projected_solid_angle"
    clipped = technique in ("clipped_solid_angle", "projected_solid_angle", "bilinear_cosine_warp_clipping_hart",
                            "biquadratic_cosine_warp_clipping_hart", "projected_solid_angle_arvo")
    vmax, vmin = v["max_light_vertices"], v.get("min_light_vertices", v["max_light_vertices"])
    d = {
        "MATERIAL_COUNT": v.get("materials", 3), "POLYGONAL_LIGHT_COUNT": v["lights"], "POLYGONAL_LIGHT_ARRAY_SIZE": max(v["lights"], 1),
        "POLYGONAL_LIGHT_COUNT_CLAMPED": min(v["lights"], 33), "LIGHT_TEXTURE_COUNT": 4,
        "MIN_POLYGON_VERTEX_COUNT_BEFORE_CLIPPING": vmin, "MAX_POLYGONAL_LIGHT_VERTEX_COUNT": vmax,
        "MAX_POLYGON_VERTEX_COUNT": vmax + (1 if clipped else 0),
        "SAMPLE_COUNT": v["samples"], "SAMPLE_COUNT_CLAMPED": min(v["samples"], 33),
        "TRACE_SHADOW_RAYS": int(v.get("rays", False)), "SHOW_POLYGONAL_LIGHTS": int(v.get("show_lights", False)),
        # error_display_t of the reference (main.h:93-118, main.c:728-750): 1..3 diffuse, 4..6 specular
        "ERROR_DISPLAY_DIFFUSE": int(1 <= v.get("error_display", 0) <= 3), "ERROR_DISPLAY_SPECULAR": int(v.get("error_display", 0) >= 4),
        "ERROR_INDEX": (v.get("error_display", 0) - 1) % 3 if v.get("error_display", 0) else 0,
        "OUTPUT_LINEAR_RGB": int(v.get("output_linear_rgb", True)),
    }
    for i, s in enumerate(strategies):
        d["SAMPLING_STRATEGIES_" + s] = int(v["strategy"] == i)
    for i, h in enumerate(heuristics):
        d["MIS_HEURISTIC_" + h] = int(v.get("heuristic", 0) == i)
    for key, name in techniques.items():
        d["SAMPLE_POLYGON_" + name] = int(key == technique)
    flags = ["-D%s=%s" % kv for kv in d.items()]
    flags.append("-DUSE_BIASED_PROJECTED_SOLID_ANGLE_SAMPLING" if biased else "-DDONT_USE_BIASED_PROJECTED_SOLID_ANGLE_SAMPLING")
    return flags


def variant_name(v):
    name = "s%d_h%d_%s_L%d_V%d-%d_S%d_r%d_l%d_o%d" % (
        v["strategy"], v.get("heuristic", 0), v.get("technique", "projected_solid_angle"), v["lights"],
        v.get("min_light_vertices", v["max_light_vertices"]), v["max_light_vertices"], v["samples"],
        int(v.get("rays", False)), int(v.get("show_lights", False)), int(v.get("output_linear_rgb", True)))
    return name + ("_e%d" % v["error_display"] if v.get("error_display", 0) else "")


# Variants that the golden fixtures and the oracle-vs-reference tests use.
# strategy / heuristic numbers follow the reference enums (src/main.h:45-89).
VARIANTS = [
    # BASELINE config 1: diffuse only, one triangle
    dict(strategy=0, lights=1, max_light_vertices=3, samples=1),
    # config 2: GGX MIS with a pentagon and shadow rays
    dict(strategy=1, heuristic=0, lights=1, max_light_vertices=5, samples=1, rays=True),
    dict(strategy=1, heuristic=1, lights=1, max_light_vertices=5, samples=2),
    # config 3 (at 2 spp to keep the fixture small): four quads, clamped optimal MIS, rays
    dict(strategy=3, heuristic=3, lights=4, max_light_vertices=4, samples=2, rays=True),
    # every MIS heuristic and the remaining strategies on mixed 3..6-gons
    dict(strategy=3, heuristic=0, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=3, heuristic=1, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=3, heuristic=2, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=3, heuristic=4, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=2, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=4, lights=3, min_light_vertices=3, max_light_vertices=6, samples=2),
    # 7-gon (largest polygon the generated clipper covers), lights visible
    dict(strategy=3, heuristic=3, lights=1, max_light_vertices=7, samples=1, show_lights=True),
    # textured lights with the light display: every texturing technique of get_polygon_radiance
    dict(strategy=3, heuristic=3, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1, show_lights=True),
    # other techniques
    dict(strategy=0, technique="projected_solid_angle_biased", lights=1, max_light_vertices=4, samples=1),
    dict(strategy=0, technique="solid_angle", lights=1, max_light_vertices=4, samples=1),
    dict(strategy=1, heuristic=0, technique="clipped_solid_angle", lights=1, max_light_vertices=4, samples=1),
    # the two simplest related-work techniques (run time baseline, Turk's area sampling)
    dict(strategy=0, technique="baseline", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2),
    dict(strategy=0, technique="area_turk", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2, rays=True),
    # Urena (a unit-square light), Arvo's spherical triangles, Hart's bilinear cosine warp
    dict(strategy=1, heuristic=0, technique="rectangle_solid_angle_urena", lights=1, max_light_vertices=4, samples=2),
    dict(strategy=0, technique="solid_angle_arvo", lights=3, min_light_vertices=3, max_light_vertices=6, samples=1),
    dict(strategy=1, heuristic=1, technique="solid_angle_arvo", lights=1, max_light_vertices=5, samples=2),
    dict(strategy=0, technique="bilinear_cosine_warp_hart", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2),
    dict(strategy=0, technique="bilinear_cosine_warp_clipping_hart", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2, rays=True),
    # Arvo's projected solid angle sampling: diffuse only, GGX MIS (the tail uses the density without the cosine), error display
    dict(strategy=0, technique="projected_solid_angle_arvo", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2),
    dict(strategy=1, heuristic=0, technique="projected_solid_angle_arvo", lights=1, max_light_vertices=5, samples=1, rays=True),
    dict(strategy=0, technique="projected_solid_angle_arvo", lights=3, min_light_vertices=3, max_light_vertices=6, samples=1, error_display=1),
    dict(strategy=0, technique="biquadratic_cosine_warp_hart", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2),
    dict(strategy=0, technique="biquadratic_cosine_warp_clipping_hart", lights=3, min_light_vertices=3, max_light_vertices=6, samples=2, rays=True),
    # error display: backward (diffuse-only path), backward times PSA and forward (combined path)
    dict(strategy=0, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1, error_display=1),
    dict(strategy=3, heuristic=3, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1, error_display=2),
    dict(strategy=3, heuristic=3, lights=3, min_light_vertices=3, max_light_vertices=6, samples=1, error_display=6),
    # encoded output (sRGB transfer in the shader)
    dict(strategy=0, lights=1, max_light_vertices=3, samples=1, output_linear_rgb=False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "_ref"))
    ap.add_argument("--only", default=None, help="substring filter on variant names")
    args = ap.parse_args()
    shader_dir = os.path.join(args.reference, "src", "shaders")
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    oracle_dir = os.path.abspath(os.path.join(HERE, ".."))
    tmp = tempfile.mkdtemp(prefix="ref_shaders_")
    try:
        for name in SHADER_FILES:
            with open(os.path.join(shader_dir, name)) as f:
                text = f.read()
            with open(os.path.join(tmp, name), "w") as f:
                f.write(translate(name, text))
        jobs = []
        for v in VARIANTS:
            name = variant_name(v)
            if args.only and args.only not in name:
                continue
            target = os.path.join(out_dir, "libref_shader_%s.so" % name)
            cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fPIC", "-shared", "-w",
                   "-I", HERE, "-I", tmp, '-DREF_SHADER_SOURCE="%s"' % os.path.join(tmp, "shading_pass.frag.glsl")]
            cmd += variant_defines(v)
            cmd += [os.path.join(HERE, "ref_shader_main.cpp"), "-o", target,
                    "-L", oracle_dir, "-loracle", "-Wl,-rpath,$ORIGIN/.."]
            jobs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            while sum(1 for _, p in jobs if p.poll() is None) >= (os.cpu_count() or 2):
                jobs[0][1].wait()
        failed = 0
        for name, p in jobs:
            out, _ = p.communicate()
            if p.returncode:
                failed += 1
                sys.stderr.write("variant %s failed:\n%s\n" % (name, out.decode()[:6000]))
        print("built %d reference shader variants into %s (%d failed)" % (len(jobs) - failed, out_dir, failed))
        return 1 if failed else 0
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
