"""CPU: include/radix_sort.hpp keeps the reference's compile-time contract (accepted / rejected types)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compiles(body):
    src = '#include "radix_sort.hpp"\n#include <string>\n#include <vector>\nstruct S { std::string key; long long data; };\n' \
          'struct P { double x; int y; };\nvoid f() {\n' + body + '\n}\n'
    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write(src)
    try:
        r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), f.name],
                           capture_output=True, text=True)
        return r.returncode == 0, r.stderr
    finally:
        os.unlink(f.name)


def test_accepted_instantiations_compile():
    ok, err = _compiles("""
        std::vector<unsigned> a; radix_sort::sort(a); radix_sort::sort(a, {}, true);
        std::vector<long long> b; radix_sort::sort(b);
        std::vector<double> c; radix_sort::sort(c, {}, true);
        std::vector<char> d; radix_sort::sort(d);
        std::vector<std::string> e; radix_sort::sort(e);
        std::vector<S> s; radix_sort::sort(s, &S::key, true); radix_sort::sort(s, &S::data);
        std::vector<P> p; radix_sort::sort(p, [](const P& q) -> const double& { return q.x; });
    """)
    assert ok, err


def test_unsupported_types_fail_with_the_reference_messages():
    ok, err = _compiles("std::vector<P> p; radix_sort::sort(p);")
    assert not ok and "Unsupported type" in err
    ok, err = _compiles("std::vector<P> p; radix_sort::sort(p, [](const P& q) { return q; });")
    assert not ok and "Key extractor is invalid" in err
