"""What MoonRTX touches on `TkOptiX._root` / `TkOptiX._canvas`, without a display.

The reference schedules work on Tk's timer queue through the renderer object: auto-advance ticks and the restore of converged
rendering after an interactive burst (`rt._root.after / after_cancel`, moon_renderer.py:398-404, :467-480), the FOV overlay's
refresh (renderer_fov.py:199-209), and -- per exported video frame and at the end of an export -- the GUI updates of
renderer_video.py:213, :361-363 (`rt._root.after(0, ...)`, `rt._root.after(500, finish)`).  With `_root = None` the first two
switch themselves off (the reference tests for None), the video export raises.  There is no Tk on a headless GPU node (no
`tkinter` in this image), so `HeadlessRoot` serves the timer queue from one thread of its own -- "the Tk main thread" of the
reference's comments: callbacks run one at a time, in due order, never concurrently -- and answers the few window calls
(`title`, `state`, `bind`, `winfo_*`, renderer_status.py:245-246, :436; renderer_dialogs.py:1621-1623) inertly.
`HeadlessCanvas` does the same for the drawing calls of the measuring line (renderer_navigation.py:631-682) and the mouse-wheel
binding (renderer_status.py:428).  Dialog windows (`tk.Toplevel(rt._root)`) stay impossible: they need Tk itself.
"""
import heapq
import itertools
import sys
import threading
import time
import traceback


class HeadlessRoot:
    def __init__(self, width=0, height=0):
        self._w, self._h = int(width), int(height)
        self._cv = threading.Condition()
        self._heap = []                    # (due time, sequence number, id)
        self._jobs = {}                    # id -> (func, args)
        self._seq = itertools.count(1)
        self._alive = True
        self._title = ""
        self._state = "normal"
        self.bindings = {}                 # sequence -> handler (a headless driver may call them: fire())
        self._thread = threading.Thread(target=self._loop, name="moonrt-ui", daemon=True)
        self._thread.start()

    # ---- Tk's timer queue
    def after(self, ms, func=None, *args):
        """Tk semantics: without `func` sleep for `ms`; else run func(*args) on the UI thread after `ms` milliseconds, return an id."""
        if func is None:
            time.sleep(max(0, ms) / 1000.0)
            return None
        with self._cv:
            if not self._alive:
                return None
            n = next(self._seq)
            ident = f"after#{n}"
            self._jobs[ident] = (func, args)
            heapq.heappush(self._heap, (time.monotonic() + max(0, ms) / 1000.0, n, ident))
            self._cv.notify_all()
            return ident

    def after_idle(self, func, *args):
        return self.after(0, func, *args)

    def after_cancel(self, ident):
        with self._cv:
            self._jobs.pop(ident, None)    # the heap entry stays and is skipped when it comes up

    def _loop(self):
        while True:
            with self._cv:
                while self._alive and (not self._heap or self._heap[0][0] > time.monotonic()):
                    self._cv.wait(None if not self._heap else max(0.0, self._heap[0][0] - time.monotonic()))
                if not self._alive:
                    return
                _, _, ident = heapq.heappop(self._heap)
                job = self._jobs.pop(ident, None)
            if job is not None:
                try:
                    job[0](*job[1])
                except Exception:          # Tk reports and carries on (Tk.report_callback_exception)
                    print("Exception in headless UI callback", file=sys.stderr)
                    traceback.print_exc()

    def pending(self):
        with self._cv:
            return len(self._jobs)

    def wait_idle(self, timeout=10.0):
        """Headless helper: block until every callback due by now has run (not the ones scheduled for later)."""
        t_end = time.monotonic() + timeout
        done = threading.Event()
        if self.after(0, done.set) is None:
            return True
        return done.wait(max(0.0, t_end - time.monotonic()))

    # ---- main loop / life cycle
    def mainloop(self, n=0):
        self._thread.join()

    def quit(self):
        self.destroy()

    def destroy(self):
        with self._cv:
            self._alive = False
            self._jobs.clear()
            self._cv.notify_all()
        if self._thread is not threading.current_thread():
            self._thread.join(timeout=5.0)

    def update(self):
        pass

    update_idletasks = update

    # ---- window calls, inert
    def title(self, string=None):
        if string is None:
            return self._title
        self._title = str(string)

    def state(self, newstate=None):
        if newstate is None:
            return self._state
        self._state = str(newstate)

    def bind(self, sequence=None, func=None, add=None):
        if func is not None:
            self.bindings[sequence] = func
        return f"bind#{len(self.bindings)}"

    def unbind(self, sequence, funcid=None):
        self.bindings.pop(sequence, None)

    def fire(self, sequence, event=None):
        """Headless driver: call the handler bound to `sequence` on the UI thread."""
        h = self.bindings.get(sequence)
        return self.after(0, h, event) if h is not None else None

    def protocol(self, name=None, func=None):
        if func is not None:
            self.bindings[name] = func

    def winfo_x(self):
        return 0

    winfo_y = winfo_rootx = winfo_rooty = winfo_x

    def winfo_width(self):
        return self._w

    def winfo_height(self):
        return self._h

    winfo_reqwidth = winfo_screenwidth = winfo_width
    winfo_reqheight = winfo_screenheight = winfo_height

    def winfo_exists(self):
        return 1 if self._alive else 0

    def geometry(self, new=None):
        if new is None:
            return f"{self._w}x{self._h}+0+0"

    def focus_set(self):
        pass

    focus_force = lift = deiconify = withdraw = focus_set


class HeadlessCanvas:
    """Item bookkeeping only: ids, coordinates and options of what the reference draws over the image (the measuring line)."""

    def __init__(self, width=0, height=0):
        self._w, self._h = int(width), int(height)
        self.items = {}
        self.bindings = {}
        self._ids = itertools.count(1)

    def _create(self, kind, coords, kw):
        n = next(self._ids)
        self.items[n] = {"type": kind, "coords": [float(c) for c in coords], "options": dict(kw)}
        return n

    def create_line(self, *coords, **kw):
        return self._create("line", coords[0] if len(coords) == 1 and hasattr(coords[0], "__len__") else coords, kw)

    def create_text(self, *coords, **kw):
        return self._create("text", coords, kw)

    def create_image(self, *coords, **kw):
        return self._create("image", coords, kw)

    def coords(self, item, *coords):
        if item not in self.items:
            return []
        if coords:
            self.items[item]["coords"] = [float(c) for c in (coords[0] if len(coords) == 1 and hasattr(coords[0], "__len__") else coords)]
        return list(self.items[item]["coords"])

    def itemconfig(self, item, **kw):
        if item in self.items:
            self.items[item]["options"].update(kw)

    itemconfigure = itemconfig

    def delete(self, *items):
        for it in items:
            if it == "all":
                self.items.clear()
            else:
                self.items.pop(it, None)

    def bind(self, sequence=None, func=None, add=None):
        if func is not None:
            self.bindings[sequence] = func
        return f"bind#{len(self.bindings)}"

    def unbind(self, sequence, funcid=None):
        self.bindings.pop(sequence, None)

    def winfo_width(self):
        return self._w

    def winfo_height(self):
        return self._h

    def focus_set(self):
        pass

    def config(self, **kw):
        pass

    configure = config
