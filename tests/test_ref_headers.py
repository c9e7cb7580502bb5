"""CPU: the three plug-in classes compile against the reference's REAL base classes (SURVEY §8 rows a16,
b, f3): `g++ -fsyntax-only -DPFHIP_WITH_FUNASR` on paraformer_hip.cpp / fsmn_vad_hip.cpp / ct_transformer_hip.cpp and on
ref_seam_check.cpp (static_asserts: derives from funasr::Model + funasr::WfstDecodable / VadModel / PuncModel, not abstract,
the overrides have the base's signatures; the calls offline-stream.cpp, tpass-stream.cpp and funasrruntime.cpp make through
the base pointers resolve).

Headers come from /root/reference where they lie (model.h, wfst-decodable.h, decoder.h, vocab.h, vad-model.h, punc-model.h,
openfst, glog's checked-in logging.h, nlohmann json).  The one GENERATED header openfst wants, gflags/gflags.h, is produced
out of tree by configuring the reference's vendored gflags with its own CMakeLists into build/ref_gen/ (a compile check
of this repo's code, not an oracle: nothing is executed or linked).  Skipped where /root/reference does not exist."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
ORT = os.path.join(REF, "onnxruntime")
HOST = os.path.join(ROOT, "asr-2pass_amd", "csrc", "host")
GEN = os.path.join(ROOT, "build", "ref_gen", "gflags")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(ORT, "include")), reason="/root/reference not present")


@pytest.fixture(scope="module")
def include_flags():
    if not os.path.exists(os.path.join(GEN, "include", "gflags", "gflags.h")):
        if shutil.which("cmake") is None:
            pytest.skip("cmake not available to generate gflags/gflags.h")
        os.makedirs(GEN, exist_ok=True)
        r = subprocess.run(["cmake", "-S", os.path.join(ORT, "third_party", "gflags"), "-B", GEN,
                            "-DCMAKE_POLICY_VERSION_MINIMUM=3.5", "-DBUILD_SHARED_LIBS=OFF"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    dirs = [os.path.join(ORT, "include"), os.path.join(ORT, "src"), os.path.join(ORT, "third_party", "openfst", "src", "include"),
            os.path.join(REF, "websocket", "third_party", "json", "include"), os.path.join(GEN, "include"),
            os.path.join(ORT, "third_party", "glog", "src")]
    return ["-I" + d for d in dirs]


@pytest.mark.parametrize("src", ["paraformer_hip.cpp", "fsmn_vad_hip.cpp", "ct_transformer_hip.cpp", "ref_seam_check.cpp"])
def test_adapter_compiles_against_reference_headers(include_flags, src):
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror=overloaded-virtual", "-DPFHIP_WITH_FUNASR"] + include_flags + [os.path.join(HOST, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_check_really_sees_the_reference_base_classes(include_flags, tmp_path):
    """Negative control: an adapter whose Forward drifted from model.h:31-32 (int* -> long*) must be rejected."""
    bad = tmp_path / "drift.cpp"
    bad.write_text('#include "model.h"\n'
                   "struct Drift : funasr::Model {\n"
                   "  void StartUtterance() override {} void EndUtterance() override {} void Reset() override {}\n"
                   "  std::string Rescoring() override { return \"\"; } int GetAsrSampleRate() override { return 16000; }\n"
                   "  std::vector<std::string> Forward(float** din, long* len, bool fin, const std::vector<std::vector<float>>& hw,\n"
                   "                                   void* dec, int batch_in) override { return {}; }\n"
                   "};\n")
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only"] + include_flags + [str(bad)], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "override" in r.stderr
