"""Run a script of this repository with a native-stack dump on SIGSEGV / SIGBUS / SIGABRT.

usage: python tools/debug/with_backtrace.py bench.py [its arguments]
(build the helper first: gcc -shared -fPIC -O1 -o tools/debug/libsegvbt.so tools/debug/segv_backtrace.c)"""
import ctypes
import os
import runpy
import sys

here = os.path.dirname(os.path.abspath(__file__))
ctypes.CDLL(os.path.join(here, 'libsegvbt.so')).segvbt_install()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name='__main__')
