"""A rendezvous port for the multi-process tests: asked from the kernel instead of hard-coded, so that
back-to-back or parallel runs on one box do not collide."""
import socket


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
