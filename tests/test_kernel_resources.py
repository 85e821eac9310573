"""Register and LDS budgets the update pipeline is built on, checked from the compiler's own
metadata (hipcc cross-compiles for gfx950 without a GPU; ~1 minute).

* The wavefront voice kernels must not spill: scratch traffic was 11 MB of HBM writes per launch
  when they did, and every variant has to stay at two workgroups per CU.
* The post-stream reduction has to FIT BESIDE two resident workgroups of the HRTF voice kernel on
  a CU: 512 VGPRs per SIMD lane and 160 KB of LDS per CU, allocation granules of
  8 registers; one wavefront of each workgroup per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openal-soft_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine",
         "-O3", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "--cuda-device-only", "-S"]

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")


def kernel_metadata(tmp_path, source, extra=()):
    out = tmp_path / (source + ".s")
    subprocess.run([HIPCC, *FLAGS, *extra, "-o", str(out), os.path.join(CSRC, source)], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    meta = {}
    # one YAML map per kernel in the amdhsa.kernels note: fields come in alphabetical order
    for block in re.split(r"\n  - ", text[text.index("amdhsa.kernels:"):])[1:]:
        fields = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
        if "name" in fields:
            meta[fields["name"]] = {k: int(v) for k, v in fields.items() if v.isdigit()}
    return meta


def makefile_flags():
    """HIPFLAGS and the per-file FLAGS_* of openal-soft_amd/Makefile, as the library is really built"""
    text = open(os.path.join(ROOT, "openal-soft_amd", "Makefile")).read()
    cxx = re.search(r"^CXXFLAGS := (.*)$", text, re.M).group(1).split()
    hip = re.search(r"^HIPFLAGS := (.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").replace("$(CXXFLAGS)", " ".join(cxx)).split()
    per_file = {m.group(1): m.group(2).split() for m in re.finditer(r"^FLAGS_(\w+)\s*:= (.*)$", text, re.M)}
    return hip, per_file


def granule(n, g=8):
    return (n + g - 1) // g * g


@pytest.fixture(scope="module")
def voice_wave(tmp_path_factory):
    # with the Makefile's own flags for the file: the voice kernels are built without machine-level loop-invariant code motion
    # (14-16 registers less per lane in every variant at the same kernel time, profiles/r5/licm_ab.txt)
    _, per_file = makefile_flags()
    assert "-disable-machine-licm" in per_file["voice_wave"]
    return kernel_metadata(tmp_path_factory.mktemp("kres"), "voice_wave.hip", per_file["voice_wave"])


@pytest.fixture(scope="module")
def voice_kernel(tmp_path_factory):
    return kernel_metadata(tmp_path_factory.mktemp("kres2"), "voice_kernel.hip")


def test_wavefront_voice_kernels_do_not_spill(voice_wave):
    waves = {n: m for n, m in voice_wave.items() if "VoiceWaveKernel" in n}
    # 6 packed-VALU variants + the two matrix-pipe FIR ones + the two with line accumulators in registers (dry lines; HRTF + one
    # slot's wet lines) + the measurement (PROF) builds of seven of them
    assert len(waves) == 17, sorted(voice_wave)
    for name, m in waves.items():
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
        assert m["vgpr_count"] + m.get("agpr_count", 0) <= 256, (name, m)   # two wavefronts per SIMD (one unified file)
        assert 2 * m["group_segment_fixed_size"] <= 160 * 1024, (name, m)   # two workgroups per CU


def test_slice_voice_kernel_keeps_its_lines_in_registers(tmp_path):
    """OALGPU_CTX_SLICE_LINES (csrc/voice_slice.hip): 24 lines x 4 frames per lane beside the resampler, no scratch, two
    workgroups per CU."""
    _, per_file = makefile_flags()
    assert "-disable-machine-licm" in per_file["voice_slice"]
    meta = kernel_metadata(tmp_path, "voice_slice.hip", per_file["voice_slice"])
    (name, m), = [(n, m) for n, m in meta.items() if "VoiceSliceKernel" in n]
    assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
    assert m["vgpr_count"] + m.get("agpr_count", 0) <= 256, (name, m)
    assert 2 * granule(m["group_segment_fixed_size"], 1280) <= 160 * 1024, (name, m)


def full_metadata(tmp_path, source):
    """the kernels of a translation unit that needs the Makefile's complete flag set (include paths, per-file flags)"""
    hip, per_file = makefile_flags()
    out = tmp_path / (source + ".s")
    subprocess.run([HIPCC, *hip, *per_file.get(source[:-4], []), "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(CSRC, source)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    meta = {}
    for block in re.split(r"\n  - ", text[text.index("amdhsa.kernels:"):])[1:]:
        fields = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
        if "name" in fields:
            meta[fields["name"]] = {k: int(v) for k, v in fields.items() if v.isdigit()}
    return meta, text


def test_voice_per_wavefront_hrtf_kernel_runs_four_wavefronts_per_simd(voice_kernel, tmp_path):
    """csrc/voice_wave16.hip (the HRTF voice kernel of round 6): a voice per wavefront at FOUR wavefronts per SIMD --
    <= 112 registers beside the post stream's reduction (512 per SIMD lane, granules of 8), no scratch, the 16-wavefront
    workgroup's LDS within one CU's 160 KB together with the post-process's granules, the 8-wavefront one twice per CU."""
    _, per_file = makefile_flags()
    assert "-disable-machine-licm" in per_file["voice_wave16"]
    meta, _ = full_metadata(tmp_path, "voice_wave16.hip")
    k16 = next(m for n, m in meta.items() if "VoiceWave16KernelILb0ELi16ELb0E" in n)
    k8 = next(m for n, m in meta.items() if "VoiceWave16KernelILb0ELi8ELb0E" in n)
    k4 = next(m for n, m in meta.items() if "VoiceWave16KernelILb0ELi4ELb0E" in n)
    k16s = next(m for n, m in meta.items() if "VoiceWave16KernelILb0ELi16ELb1E" in n)       # with sends (stream rows out of the registers)
    reduce4 = next(m for n, m in voice_kernel.items() if "BusReduceKernelILi4E" in n)
    for m in (k16, k8):
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, m
        assert 4 * granule(m["vgpr_count"] + m.get("agpr_count", 0)) <= 512, m        # four wavefronts per SIMD
    # the headline form (16 wavefronts: a machine-filling scene): four voice wavefronts AND one of the reduction's on a SIMD
    assert k16["vgpr_count"] + k16.get("agpr_count", 0) <= 112, k16
    # ... and the same with sends (BASELINE configs[4]): read back from LDS the send's filter states put it at 123 registers,
    # and the convolution's chain behind it lost its place beside the voices
    assert k16s["vgpr_spill_count"] == 0 and k16s["private_segment_fixed_size"] == 0 and k16s["vgpr_count"] + k16s.get("agpr_count", 0) <= 112, k16s
    assert 4 * granule(k16["vgpr_count"] + k16.get("agpr_count", 0)) + granule(reduce4["vgpr_count"]) <= 512, (k16, reduce4)
    post = kernel_metadata(tmp_path, "post_wave.hip", ["-mllvm", "-amdgpu-load-store-vectorizer=0"])
    fused = next(m for n, m in post.items() if "PostFusedKernel" in n)
    assert granule(k16["group_segment_fixed_size"], 1280) + granule(fused["group_segment_fixed_size"], 1280) <= 160 * 1024, k16
    assert 2 * granule(k8["group_segment_fixed_size"], 1280) <= 160 * 1024, k8
    # the resident launch of the 16-wavefront form (OALGPU_CTX_RESIDENT): the voice workgroup never leaves its CU, and the reduction and
    # the post-process of every update WAIT for it -- one wavefront of theirs must fit on a SIMD beside four voice wavefronts, one
    # workgroup of each in the LDS the voice workgroup leaves (or nothing ever moves again)
    k16r = next(m for n, m in meta.items() if "VoiceWave16KernelILb0ELi16ELb0ELb1E" in n)
    assert k16r["vgpr_spill_count"] == 0 and k16r["private_segment_fixed_size"] == 0, k16r
    red = next(m for n, m in post.items() if "BusReduceResidentKernel" in n)
    pst = next(m for n, m in post.items() if "PostResidentKernel" in n)
    for other in (red, pst):
        assert other["vgpr_spill_count"] == 0, other
        assert 4 * granule(k16r["vgpr_count"] + k16r.get("agpr_count", 0)) + granule(other["vgpr_count"]) <= 512, (k16r, other)
    assert granule(k16r["group_segment_fixed_size"], 1280) + granule(red["group_segment_fixed_size"], 1280) + granule(pst["group_segment_fixed_size"], 1280) <= 160 * 1024, k16r
    # (small scenes: four wavefronts per workgroup, two per SIMD at most -- the register form of rounds 1-5)
    assert k4["vgpr_spill_count"] == 0 and k4["vgpr_count"] + k4.get("agpr_count", 0) <= 256, k4


def test_rows_in_lds_kernel_keeps_32_lines_per_wavefront_in_registers(tmp_path):
    """csrc/voice_rows.hip (dry lines AND sends): eight wavefronts, two per SIMD, 32 line accumulators x 2 frames per
    lane in registers for the whole launch -- no scratch (an accumulator indexed at run time, or a whole-array copy of the
    resampler's outputs, puts them there: both happened while it was written) -- and ONE workgroup per CU."""
    _, per_file = makefile_flags()
    assert "-disable-machine-licm" in per_file["voice_rows"]
    meta, _ = full_metadata(tmp_path, "voice_rows.hip")
    (name, m), = [(n, m) for n, m in meta.items() if "VoiceRowsKernelILb0E" in n]
    assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, (name, m)
    assert m["vgpr_count"] + m.get("agpr_count", 0) <= 256, (name, m)
    assert 80 * 1024 < m["group_segment_fixed_size"] <= 160 * 1024, (name, m)      # (one per CU by its LDS: the grid is sized for that)


def test_reduction_fits_beside_the_hrtf_voice_kernel(voice_wave, voice_kernel, tmp_path):
    reduce4 = next(m for n, m in voice_kernel.items() if "BusReduceKernelILi4E" in n)
    assert reduce4["vgpr_spill_count"] == 0
    # the one-launch HRTF post-process runs beside the voice kernel too (one wavefront per workgroup)
    post = kernel_metadata(tmp_path, "post_wave.hip", ["-mllvm", "-amdgpu-load-store-vectorizer=0"])
    fused = next(m for n, m in post.items() if "PostFusedKernel" in n)
    assert fused["vgpr_spill_count"] == 0 and fused["vgpr_count"] <= reduce4["vgpr_count"]
    # four LDS granules: with seven (a second array for four channels' decoder taps) the post-process no longer started beside a
    # resident voice kernel and its polling reduction (round 5)
    assert fused["group_segment_fixed_size"] <= 4 * 1280
    for variant in ("VoiceWaveKernelILi17ELi64ELi0ELb0ELb0ELb0E", "VoiceWaveKernelILi17ELi64ELi0ELb0ELb1ELb0E"):   # VALU / matrix-pipe FIR
        voice = next(m for n, m in voice_wave.items() if variant in n)
        # per SIMD lane: one wavefront of each of the two voice workgroups + one of the reduction's four
        assert 2 * granule(voice["vgpr_count"]) + granule(reduce4["vgpr_count"]) <= 512, (voice, reduce4)
        # LDS is allocated in granules of 1280 bytes on gfx950
        assert 2 * granule(voice["group_segment_fixed_size"], 1280) + granule(reduce4["group_segment_fixed_size"], 1280) <= 160 * 1024


def test_resident_voice_kernel_leaves_room_for_what_waits_for_it(voice_wave, tmp_path):
    """OALGPU_CTX_RESIDENT: the voice kernel never leaves the machine, and the reduction and the post-process of every update
    WAIT for it (device counters) -- they must be able to start on a SIMD that holds one wavefront of each of a CU's two
    resident voice workgroups, or nothing ever moves again.  The resident kernel is its own translation unit with the
    Makefile's flags for it (machine LICM off: 217 registers instead of 249)."""
    hip, per_file = makefile_flags()
    assert "-disable-machine-licm" in per_file["voice_wave_res"]
    out = tmp_path / "voice_wave_res.s"
    subprocess.run([HIPCC, *hip, *per_file["voice_wave_res"], "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", str(out),
                    os.path.join(CSRC, "voice_wave_res.hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    meta = {}
    for block in re.split(r"\n  - ", text[text.index("amdhsa.kernels:"):])[1:]:
        fields = dict(re.findall(r"\.(\w+):\s+(\S+)", block))
        if "name" in fields:
            meta[fields["name"]] = {k: int(v) for k, v in fields.items() if v.isdigit()}
    assert len(meta) == 1, sorted(meta)
    voice = next(iter(meta.values()))
    assert voice["vgpr_spill_count"] == 0 and voice["private_segment_fixed_size"] == 0, voice
    post = kernel_metadata(tmp_path, "post_wave.hip", ["-mllvm", "-amdgpu-load-store-vectorizer=0"])
    red = next(m for n, m in post.items() if "BusReduceResidentKernel" in n)
    pst = next(m for n, m in post.items() if "PostResidentKernel" in n)
    for other in (red, pst):
        assert other["vgpr_spill_count"] == 0, other
        assert 2 * granule(voice["vgpr_count"]) + granule(other["vgpr_count"]) <= 512, (voice, other)
    lds = 2 * granule(voice["group_segment_fixed_size"], 1280) + granule(red["group_segment_fixed_size"], 1280) + granule(pst["group_segment_fixed_size"], 1280)
    assert lds <= 160 * 1024, lds
    # the launched product kernel's LDS footprint, which the resident one shares
    launched = next(m for n, m in voice_wave.items() if "VoiceWaveKernelILi17ELi64ELi0ELb0ELb1ELb0E" in n)
    assert launched["group_segment_fixed_size"] == voice["group_segment_fixed_size"]


def test_no_packed_fp32_op_takes_src0_low_and_src1_high():
    """gfx950: v_pk_{fma,mul,add}_f32 with op_sel = [0,1,..] (low lane = src0.lo x src1.HI) reads src1.hi as zero in
    a few percent of its executions while another wavefront of the SIMD executes MFMAs (tools/ubench_pk_opsel.hip,
    DESIGN.md 3.9) -- and every kernel here may run beside the matrix-pipe FIR of the HRTF voice kernel.  The Makefile
    keeps the passes that create the form switched off; this checks the ISA of every kernel, built with the
    Makefile's own flags."""
    import tempfile
    hip, per_file = makefile_flags()
    assert "-fno-slp-vectorize" in hip and "-disable-vector-combine" in hip
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(os.listdir(CSRC)):
            if not src.endswith(".hip"):
                continue
            out = os.path.join(tmp, src + ".s")
            subprocess.run([HIPCC, *hip, *per_file.get(src[:-4], []), "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                            os.path.join(CSRC, src)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for line in open(out):
                if re.search(r"v_pk_\w+_f32 .*op_sel:\[0,1", line):
                    bad.append((src, line.strip()))
    assert not bad, bad[:10]
