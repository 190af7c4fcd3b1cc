"""Minimal stream-capture patterns (torch only, no bmhrl kernels) to find which fork / join shape makes
hipStreamEndCapture / hipGraphLaunch crash on this ROCm build.  One pattern per process:

    python tests/probes/capture_forks.py <pattern>

patterns: flat (A forked from main twice, one after the other), nested (B forked from A inside A's fork, joined to A, A joined
to main), nested_cross (B forked from A, joined straight to main), two_sides (A and B both forked from main, overlapping),
reuse (the same side stream forked 40 times), pool_wrap (40 fresh torch.cuda.Stream() objects: the pool of 32 wraps and
aliases the capture stream)."""
import sys
import torch

pat = sys.argv[1]
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
y = torch.zeros(1 << 16, device=dev)
z = torch.zeros(1 << 16, device=dev)
A, B = torch.cuda.Stream(), torch.cuda.Stream()


def say(msg):
    if torch.cuda.is_current_stream_capturing():
        print("  [capturing]", msg, flush=True)


def body():
    main = torch.cuda.current_stream()
    if pat == "flat":
        for _ in range(2):
            A.wait_stream(main)
            with torch.cuda.stream(A):
                y.add_(1)
            x.add_(1)
            main.wait_stream(A)
    elif pat in ("nested", "nested_idle"):
        A.wait_stream(main)
        with torch.cuda.stream(A):
            y.add_(1)
            B.wait_stream(A)
            with torch.cuda.stream(B):
                z.add_(1)
            if pat == "nested":
                y.add_(1)                     # A works while B works (nested_idle: A only waits for B)
            say("before A.wait_stream(B)")
            A.wait_stream(B)
            say("after A.wait_stream(B)")
            y.add_(1)
        x.add_(1)
        main.wait_stream(A)
        say("after main.wait_stream(A)")
    elif pat == "nested_cross":
        A.wait_stream(main)
        with torch.cuda.stream(A):
            y.add_(1)
            B.wait_stream(A)
            with torch.cuda.stream(B):
                z.add_(1)
            y.add_(1)
        x.add_(1)
        main.wait_stream(A)
        main.wait_stream(B)
    elif pat == "two_sides":
        A.wait_stream(main)
        B.wait_stream(main)
        with torch.cuda.stream(A):
            y.add_(1)
        with torch.cuda.stream(B):
            z.add_(1)
        x.add_(1)
        main.wait_stream(A)
        main.wait_stream(B)
    elif pat == "reuse":
        for _ in range(40):
            A.wait_stream(main)
            with torch.cuda.stream(A):
                y.add_(1)
            x.add_(1)
            main.wait_stream(A)
    elif pat == "pool_wrap":
        for _ in range(40):
            S = torch.cuda.Stream()
            S.wait_stream(main)
            with torch.cuda.stream(S):
                y.add_(1)
            x.add_(1)
            main.wait_stream(S)
    else:
        raise SystemExit("unknown pattern")


body()
torch.cuda.synchronize()
x.zero_(); y.zero_(); z.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
    print("  [capturing] body done, ending capture", flush=True)
print(pat, "captured", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(pat, "replayed", float(x[0]), float(y[0]), float(z[0]), flush=True)
