"""Drop-in replacements for the reference's `network` package (DVQ-VAE/network/):
same module paths, class names, constructor and forward signatures and state_dict
keys; the forward bodies call the HIP kernels of libdvq_hip.so."""
