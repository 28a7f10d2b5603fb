"""Empty stand-in: only --mmap touches h5py."""
