"""Build hook: the package is pure Python over one shared library, built in-tree by make/hipcc (gfx950 only)."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py


class build_with_hip(build_py):
    def run(self):
        here = os.path.dirname(os.path.abspath(__file__))
        subprocess.run(["make", "-C", os.path.join(here, "deepgrp_amd", "csrc"), "-j4"], check=True)
        super().run()


setup(cmdclass={"build_py": build_with_hip})
