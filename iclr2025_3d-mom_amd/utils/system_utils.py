"""utils/system_utils.py of the reference: directory helpers."""
import os


def mkdir_p(folder_path):
    os.makedirs(folder_path, exist_ok=True)


def searchForMaxIteration(folder):
    return max(int(fname.split("_")[-1]) for fname in os.listdir(folder))
