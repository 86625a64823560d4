"""utils/timer.py of the reference: a wall-clock timer that can be paused (train_4DGS.py pauses it around logging)."""
import time


class Timer:
    def __init__(self):
        self.start_time, self.elapsed, self.paused = None, 0, False

    def start(self):
        if self.start_time is None:
            self.start_time = time.time()
        elif self.paused:
            self.start_time, self.paused = time.time() - self.elapsed, False

    def pause(self):
        if not self.paused:
            self.elapsed, self.paused = time.time() - self.start_time, True

    def get_elapsed_time(self):
        return self.elapsed if self.paused else time.time() - self.start_time
