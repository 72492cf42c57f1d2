"""The host worker pool of liblgrasp.so (csrc/lg_pool.h) under ThreadSanitizer: jobs of growing and shrinking size run
back to back from one caller (the per-frame result path of lg_select_grasp), every index exactly once, no data race
(the first pool kept the job's bounds in the pool object and could hand a stale index to the next job)."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "leaf-grasping-vision-ml_amd", "csrc")

DRIVER = r"""
#include "lg_pool.h"
#include <cstdio>
int main() {
    LgPool pool(7);
    long long bad = 0;
    for (int round = 0; round < 3000; round++) {
        const int n = 1 + (round * 37) % 61 + (round % 5 == 0 ? 200 : 0);   // sizes jump up and down between calls
        std::vector<std::atomic<int>> hits(n);
        for (auto& h : hits) h.store(0);
        std::vector<int> results(n, -1);
        pool.run(n, [&](int i) { hits[i].fetch_add(1); results[i] = i * 3 + round; });
        for (int i = 0; i < n; i++) bad += (hits[i].load() != 1) + (results[i] != i * 3 + round);
    }
    std::printf("bad=%lld\n", bad);
    return bad != 0;
}
"""


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_pool_under_sanitizer(tmp_path, sanitizer):
    src = tmp_path / "pool_driver.cpp"
    src.write_text(DRIVER)
    exe = tmp_path / f"pool_{sanitizer}"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", f"-fsanitize={sanitizer}", "-fno-omit-frame-pointer", "-I" + CSRC,
                           str(src), "-o", str(exe), "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad=0" in r.stdout
