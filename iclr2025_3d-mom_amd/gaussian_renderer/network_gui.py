"""The SIBR viewer socket of the reference (gaussian_renderer/network_gui.py) is out of scope: this module
only keeps train_4DGS.py's polling lines (:120-146) inert -- there is never a connection."""
conn = None
addr = None


def init(wish_host, wish_port):
    return None


def try_connect():
    return None


def receive():
    raise RuntimeError("network_gui: the viewer protocol is not part of this build")


def send(message_bytes, verify):
    raise RuntimeError("network_gui: the viewer protocol is not part of this build")
