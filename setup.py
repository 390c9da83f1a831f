"""Packaging of the MI355X build: installs the `ribodetector` command of the reference (its GPU product; there is no
`ribodetector_cpu` here - the kernels need a gfx950 device) and ships the prebuilt in-tree libraries, which
`python -c "import __graft_entry__ as g; g.build()"` produces (hipcc for gfx950, g++ for the host library)."""
import os

from setuptools import find_packages, setup

here = os.path.dirname(os.path.abspath(__file__))
about = {}
with open(os.path.join(here, "ribodetector_amd", "__init__.py")) as fh:
    exec(fh.read(), about)

setup(
    name="ribodetector-mi355x",
    version=about["__version__"],
    description="RiboDetector's batched BiLSTM inference path as hand-written HIP kernels for AMD MI355X (gfx950)",
    packages=find_packages(include=["ribodetector_amd", "ribodetector_amd.*"]),
    package_data={"ribodetector_amd": ["config.json", "data/*.safetensors", "data/*.json", "csrc/*.so"]},
    python_requires=">=3.8",
    install_requires=["numpy", "torch"],
    entry_points={"console_scripts": ["ribodetector = ribodetector_amd.detect:main"]},
    zip_safe=False,
)
