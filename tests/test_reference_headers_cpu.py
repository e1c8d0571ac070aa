"""The drop-in boundary against the REAL reference headers (container only: /root/reference does not travel
to the GPU box, so this is a CPU test that skips there). The bindings INTEGRATION.md tells a maintainer to
add are taken out of INTEGRATION.md as they stand, compiled with the reference's own similarity_matrix.hpp /
expectation_maximization.hpp / util/mat.hpp / sequenced_data.hpp (C++20, as the reference builds), and
must define exactly the symbols the reference's callers link against."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "similarity_matrix.hpp")),
                                reason="the reference tree is only present in the build container")


def binding_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return [b for b in re.findall(r"```cpp\n(.*?)```", text, flags=re.S) if "(reference tree) -- MI355X binding" in b]


def compile_and_symbols(src, tmp_path, name):
    cpp = tmp_path / (name + ".cpp")
    cpp.write_text(src)
    obj = str(tmp_path / (name + ".o"))
    subprocess.run(["g++", "-std=c++20", "-O1", "-Wall", "-Wextra", "-c", str(cpp), "-o", obj, "-I" + REF,
                    "-I" + os.path.join(ROOT, "include")], check=True)
    return subprocess.run(["nm", "-C", "--defined-only", obj], check=True, capture_output=True, text=True).stdout


def test_integration_bindings_compile_against_the_reference_headers(tmp_path):
    blocks = binding_blocks()
    assert len(blocks) == 2  # computeSimilarityMatrix, expectation_maximization
    simmat = next(b for b in blocks if "computeSimilarityMatrix" in b)
    syms = compile_and_symbols(simmat, tmp_path, "similarity_matrix_mi355x")
    # the symbol divide_cluster links against (spectral_clustering.cpp:354-356; declared similarity_matrix.hpp:51-60)
    assert re.search(r" T computeSimilarityMatrix\(std::vector<std::vector<PosData.*unsigned int, unsigned int, "
                     r"std::vector<unsigned int.*double, double, double, unsigned int, std::.*string.* const&, "
                     r"std::.*string.* const&\)", syms), syms
    em = next(b for b in blocks if "expectation_maximization" in b and "computeSimilarityMatrix" not in b)
    syms = compile_and_symbols(em, tmp_path, "expectation_maximization_mi355x")
    assert re.search(r" T expectation_maximization\(std::vector<std::vector<PosData", syms), syms


def test_the_shim_writes_straight_into_the_reference_matrix(tmp_path):
    """Mat<double> has contiguous row-major storage behind data() (util/mat.hpp:238, :117): the shim must pick
    the direct path (no second N x N buffer, no element-by-element copy) for the reference's own type."""
    src = '''
#include "similarity_matrix.hpp"
#include <secedo_simmat.hpp>
static_assert(secedo_amd::detail::has_double_data<Matd>::value, "Matd::data() must be detected");
struct NoData { NoData(unsigned, unsigned); double &operator()(unsigned, unsigned); };
static_assert(!secedo_amd::detail::has_double_data<NoData>::value, "types without data() take the copy path");
int main() { return 0; }
'''
    cpp = tmp_path / "detect.cpp"
    cpp.write_text(src)
    subprocess.run(["g++", "-std=c++20", "-fsyntax-only", str(cpp), "-I" + REF, "-I" + os.path.join(ROOT, "include")],
                   check=True)
