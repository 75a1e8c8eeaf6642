"""qea — MI355X host layer: ctypes binding of libqea_hip.so, kernel schedules (engines), autograd bridges, fused Adam, data parallelism."""
