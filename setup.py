"""Packaging for the MI355X build of the `vilma fit` path: exposes the reference's console script
name (`vilma = vilma.frontend:main`, reference setup.py:20-22).  The HIP library is built in-tree
with `python -m vilma_amd.build` (hipcc, gfx950); it is a build artefact, not a wheel payload."""
import setuptools

setuptools.setup(
    name='vilma-amd',
    version='0.0.16+mi355x.1',
    description='vilma fit on AMD Instinct MI355X: hand-written HIP kernels behind the '
                'reference CLI / class API',
    packages=['vilma_amd'],
    package_data={'vilma_amd': ['libvilma_hip.so', 'csrc/*']},
    install_requires=['numpy>=1.20.0', 'pandas>=1.2.1', 'scipy', 'torch'],
    entry_points={'console_scripts': ['vilma = vilma_amd.frontend:main']},
)
