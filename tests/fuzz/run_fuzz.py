"""Runs tests/fuzz/fuzz_parsers.py in a child process with AddressSanitizer preloaded, against libnann_host_asan.so.
usage: [FUZZ_EXAMPLES=100000] python tests/fuzz/run_fuzz.py [parser ...]   -> exit status of the fuzzers; their report on stdout"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def run(parsers=(), examples=None, timeout=None):
    from nann_amd import index_build
    lib = index_build.build_host_lib(sanitize=True)
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, NANN_FUZZ_LIB=lib, LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:handle_segv=1:allocator_may_return_null=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    if examples is not None:
        env["FUZZ_EXAMPLES"] = str(examples)
    return subprocess.run([sys.executable, os.path.join(HERE, "fuzz_parsers.py")] + list(parsers), env=env, capture_output=True,
                          text=True, timeout=timeout)


if __name__ == "__main__":
    r = run(sys.argv[1:])
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr[-8000:])
    sys.exit(r.returncode)
