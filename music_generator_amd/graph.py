"""A minimal functional-graph engine with the surface of keras.Model that trainer/models.py and
trainer/task.py rely on: symbolic tensors, layers that record their calls, `Model(inputs,
outputs)` with `.layers` (input first, ordered by depth), `.input`, `.output_shape`,
`.trainable_variables`, `__call__` / `predict`, plus an explicit tape: `forward(x, save=True)`
returns the saved activations and `backward(tape, dy, ...)` walks the plan in reverse.

Execution is a flat plan of kernel launches (no tracing, no autograd).  A contraction layer
followed only by a SoftRectifier / StaticOptTanh is run as one fused kernel.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import numpy as np
import torch

from . import backend as K
from . import ops


class _ShapeOnly:
    def __init__(self, shape):
        self.shape = tuple(shape)


class KTensor:
    """Symbolic tensor: shape with a leading None batch dimension."""

    def __init__(self, shape: Tuple, node: Optional["Node"]):
        self.shape = tuple(shape)
        self.node = node
        self.fused_group = None      # FusedGroup that can produce this tensor in one kernel

    def __getitem__(self, idx):
        # old_model.input[0].shape is used by add_discriminator_block (models.py:231)
        if idx == 0:
            return _ShapeOnly(self.shape[1:])
        raise IndexError("only [0] (drop the batch axis) is supported on a symbolic tensor")


class Node:
    def __init__(self, layer: "Layer", inputs: List[KTensor]):
        self.layer = layer
        self.inputs = inputs
        self.output: Optional[KTensor] = None


class Layer:
    """Base class: `layer(symbolic)` records a call; numerical work happens in forward/backward
    which receive device tensors (batch axis first) and call the C-ABI."""

    def __init__(self, name: Optional[str] = None, **kwargs):
        self.name = name or K.unique_name(type(self).__name__)
        self._nodes: List[Node] = []
        self._built = False
        self.weights: List[K.Variable] = []
        self.trainable = True

    # -- graph construction --
    def build(self, input_shape):
        pass

    def compute_output_shape(self, input_shape):
        return input_shape

    def __call__(self, inputs):
        ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        if not all(isinstance(t, KTensor) for t in ins):
            raise TypeError("layers are called on symbolic tensors; run data through a Model")
        shapes = [t.shape for t in ins]
        in_shape = shapes if isinstance(inputs, (list, tuple)) else shapes[0]
        if not self._built:
            self.build(in_shape)
            self._built = True
        node = Node(self, ins)
        node.output = KTensor(self.compute_output_shape(in_shape), node)
        self._nodes.append(node)
        return node.output

    @property
    def output(self) -> KTensor:
        if len(self._nodes) != 1:
            raise AttributeError("layer %s has %d inbound nodes; .output is ambiguous" % (self.name, len(self._nodes)))
        return self._nodes[0].output

    @property
    def input(self):
        if len(self._nodes) != 1:
            raise AttributeError("layer %s has %d inbound nodes; .input is ambiguous" % (self.name, len(self._nodes)))
        ins = self._nodes[0].inputs
        return ins[0] if len(ins) == 1 else ins

    @property
    def trainable_weights(self) -> List[K.Variable]:
        return [v for v in self.weights if v.trainable] if self.trainable else []

    def add_weight(self, kind: str, value: np.ndarray) -> K.Variable:
        v = K.Variable("%s/%s" % (self.name, kind), value)
        self.weights.append(v)
        return v

    def get_config(self):
        return {"name": self.name}

    # -- execution (device tensors) --
    def forward(self, xs: List[torch.Tensor], save: bool):
        """returns (y, saved)"""
        raise NotImplementedError

    def backward(self, saved, dy: torch.Tensor, need: List[bool], existing: List[Optional[torch.Tensor]],
                 wgrad: bool):
        """returns a list with, per input, None or (tensor, owned)"""
        raise NotImplementedError


class InputLayer(Layer):
    def __init__(self, shape, name=None):
        super().__init__(name=name or K.unique_name("Input"))
        self.shape = tuple(shape)
        node = Node(self, [])
        node.output = KTensor((None,) + self.shape, node)
        self._nodes.append(node)
        self._built = True


def Input(shape, name=None) -> KTensor:
    """keras.layers.Input: returns the symbolic tensor of a new input layer."""
    return InputLayer(shape, name=name).output


# --------------------------------------------------------------------------------------------
class _Op:
    """One step of a model's execution plan: a node, optionally fused with the activation node
    that consumes it."""

    def __init__(self, node: Node, act_node: Optional[Node] = None):
        self.node = node
        self.act_node = act_node
        self.inputs = node.inputs
        self.output = act_node.output if act_node is not None else node.output


class FusedGroup:
    """A sub-graph (input tensor -> output tensor through `nodes`) for which fused kernels exist.
    `run(x)` returns the output tensor (forwards that keep no tape).  When `run_save` and
    `backward` are set, taped forwards use the group too: `run_save(x)` returns (y, saved) and
    `backward(saved, dy, need_weight_grads)` returns dx after adding the weight gradients of the
    group's layers into their bound gradient views."""

    # Upper bound in bytes for the block-sized scratch a group may hold from a taped forward to its
    # backward (None: no bound).  Groups that would exceed it keep only their input and re-compute.
    side_budget_bytes = None
    # without a bound the taped forward checks the free memory once its tape exceeds this many bytes
    auto_budget_from_bytes = 4 << 30
    # bytes the caller knows it will allocate between this taped forward and its backward (the train step
    # announces the discriminator tape that follows the generator's forward in the generator update): the
    # automatic bound counts them as taken
    reserve_bytes = 0
    # (tape bytes, bound chosen or None) of the automatic decisions taken so far, newest last (diagnostics)
    decisions: list = []

    def __init__(self, input: KTensor, output: KTensor, nodes: List[Node], optional_nodes: List[Node], run):
        self.input, self.output, self.nodes, self.optional_nodes, self.run = input, output, nodes, optional_nodes, run
        self.run_save = None
        self.backward = None


class Tape(list):
    """Saved tensors of one taped forward, aligned with the plan that produced them."""
    plan = None


class _FusedOp:
    def __init__(self, group: FusedGroup):
        self.group = group
        self.inputs = [group.input]
        self.output = group.output
        self.act_node = None


class Model:
    """keras.Model(inputs, outputs) for single-input single-output graphs."""

    use_fused_groups = True      # forwards may use fused sub-graph kernels
    use_fused_backward = True    # taped forwards too, where the group has a fused backward

    def __init__(self, inputs: KTensor, outputs: KTensor, name: Optional[str] = None):
        if isinstance(inputs, (list, tuple)) or isinstance(outputs, (list, tuple)):
            raise ValueError("single input / single output only")
        self.name = name or K.unique_name("Model")
        self.input = inputs
        self.output = outputs
        self._map_graph()
        self._grad_flat: Optional[torch.Tensor] = None
        self._grad_layout: Optional[List[Tuple[K.Variable, int, int]]] = None

    # -- structure --
    def _map_graph(self):
        order: List[Node] = []
        seen = set()
        layer_index: Dict[int, int] = {}

        def visit(t: KTensor):
            node = t.node
            if id(node) in seen:
                return
            seen.add(id(node))
            for it in node.inputs:
                visit(it)
            if id(node.layer) not in layer_index:
                layer_index[id(node.layer)] = len(layer_index)
            order.append(node)

        import sys
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(max(old, 10000))
        try:
            visit(self.output)
        finally:
            sys.setrecursionlimit(old)
        if id(self.input.node) not in seen:
            raise ValueError("model output does not depend on its input")
        self._nodes = order  # topological (inputs first)

        # Keras orders `.layers` by decreasing depth (distance to the output), ties by traversal.
        depth: Dict[int, int] = {}
        ldepth: Dict[int, int] = {}
        for node in reversed(order):
            d = max(depth.get(id(node), 0), ldepth.get(id(node.layer), 0))
            depth[id(node)] = d
            ldepth[id(node.layer)] = d
            for it in node.inputs:
                depth[id(it.node)] = max(depth.get(id(it.node), 0), d + 1)
        layers = {}
        for node in order:
            layers[id(node.layer)] = node.layer
        maxd = max(ldepth.values()) if ldepth else 0
        ldepth[id(self.input.node.layer)] = maxd
        self.layers: List[Layer] = sorted(layers.values(),
                                          key=lambda l: (-ldepth[id(l)], layer_index[id(l)]))

        # execution plan with conv/dense + activation fusion
        consumers: Dict[int, List[Node]] = {}
        for node in order:
            for it in node.inputs:
                consumers.setdefault(id(it), []).append(node)
        plan: List[_Op] = []
        fused = set()
        for node in order:
            if isinstance(node.layer, InputLayer) or id(node) in fused:
                continue
            act = None
            if getattr(node.layer, "fusable_activation", False) and node.output is not self.output:
                cons = consumers.get(id(node.output), [])
                if len(cons) == 1 and getattr(cons[0].layer, "is_soft_rectifier", False):
                    act = cons[0]
                    fused.add(id(act))
            plan.append(_Op(node, act))
        self._plan = plan
        self._last_use = self._compute_last_use(plan)

        # tape-free plan: replace every fusable sub-graph that lies wholly inside this model
        in_model = {id(n) for n in order}
        groups = [n.output.fused_group for n in order if n.output.fused_group is not None]
        drop, emit_at = set(), {}
        for g in groups:
            if not all(id(n) in in_model for n in g.nodes):
                continue
            inner = {id(n) for n in g.nodes}
            # an optional node (e.g. the UpSampling2D whose output also feeds the fade-in blend) is
            # dropped only when nothing outside the group consumes it in this model
            for n in g.optional_nodes:
                if id(n) in in_model and all(id(c) in inner for c in consumers.get(id(n.output), [])):
                    inner.add(id(n))
            escapes = any(id(c) not in inner for n in g.nodes if n.output is not g.output
                          for c in consumers.get(id(n.output), []))
            if escapes or any(n.output is self.output and n.output is not g.output for n in g.nodes):
                continue
            drop |= inner
            emit_at[id(g.output.node)] = g
        if emit_at:
            plan_i: List = []
            for op in plan:
                nid = id(op.node)
                if nid in emit_at:
                    plan_i.append(_FusedOp(emit_at[nid]))
                elif nid in drop or (op.act_node is not None and id(op.act_node) in drop):
                    continue
                else:
                    plan_i.append(op)
            self._plan_infer = plan_i
        else:
            self._plan_infer = plan
        self._last_use_infer = self._compute_last_use(self._plan_infer)

    def _plan_links(self, plan):
        """(index of the step that produces each tensor, number of steps that consume it) for a plan."""
        cache = self.__dict__.setdefault("_links", {})
        hit = cache.get(id(plan))
        if hit is None:
            producer = {id(op.output): i for i, op in enumerate(plan)}
            n_consumers: Dict[int, int] = {}
            for op in plan:
                for t in op.inputs:
                    n_consumers[id(t)] = n_consumers.get(id(t), 0) + 1
            hit = cache[id(plan)] = (producer, n_consumers)
        return hit

    @staticmethod
    def _compute_last_use(plan):
        last_use: Dict[int, int] = {}
        for i, op in enumerate(plan):
            for it in op.inputs:
                last_use[id(it)] = i
        return last_use

    @property
    def output_shape(self):
        return self.output.shape

    @property
    def input_shape(self):
        return self.input.shape

    @property
    def trainable_variables(self) -> List[K.Variable]:
        out, seen = [], set()
        for l in self.layers:
            for v in l.trainable_weights:
                if id(v) not in seen:
                    seen.add(id(v))
                    out.append(v)
        return out

    @property
    def trainable_weights(self):
        return self.trainable_variables

    def count_params(self) -> int:
        return sum(v.size for v in self.trainable_variables)

    def tape_bytes_estimate(self, n: int) -> int:
        """Bytes a taped forward of `n` samples keeps for its backward, from the plan's symbolic shapes: every
        step's output (the next step's saved input), in bf16 where the bf16 kernels store it, plus the saved
        activation derivative where one is kept (float32 convolutions; bf16 ones only with
        backend.BF16_SAVE_DERIVATIVE).  An estimate for memory planning, not an accounting."""
        total = 0
        bf = K.bf16_enabled()
        for op in self._plan:
            shape = op.output.shape
            elems = 1
            for d in shape[1:]:
                elems *= int(d)
            conv = getattr(op.node.layer, "fusable_activation", False) and len(shape) == 4
            half = bf and conv and shape[1] >= K.BF16_MIN_ROWS and shape[-1] in (32, 128)
            copies = 2 if (op.act_node is not None and (not half or K.BF16_SAVE_DERIVATIVE)) else 1
            total += elems * (2 if half else 4) * copies
        return int(total) * int(n)

    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise KeyError(name)

    # -- data plumbing --
    def _to_device(self, x) -> torch.Tensor:
        if isinstance(x, torch.Tensor):
            t = x
        else:
            t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        t = t.to(device=K.device(), dtype=torch.float32).contiguous()
        if tuple(t.shape[1:]) != tuple(self.input.shape[1:]):
            raise ValueError("input shape %s does not match the model input %s" % (tuple(t.shape), self.input.shape))
        return t

    # -- execution --
    # True: two consecutive convolution + activation steps of a tape-free bf16 pass run as one launch with the
    # intermediate in LDS (mgx_conv_pair_bf16; bit-identical results).  Off by default: measured 25 % SLOWER than
    # the two launches it replaces (0.61 against 2 x 0.24 ms on the stage-3 step's mix): the convolution kernels
    # are bound by instruction issue (MFMA + activation + LDS reads), not by the HBM traffic the fusion saves,
    # and the fused tile does 1.22x the work of both layers with two more barriers.
    use_conv_pairs = False
    # True: runs of two or three consecutive Conv2D(32) + SoftRectifier steps of a tape-free bf16 pass ({3x3, 3x3,
    # 2x2}, {3x3, 3x3}, {3x3, 2x2}: the groups of a discriminator block) go through the row-streaming chain kernel
    # (mgx_conv_chain_bf16; bit-identical results, the inner layers' outputs never reach HBM).  The same switch
    # covers the other launch-saving rules of tape-free passes: the from-audio + 3x3 kernel (bf16) and the fade-in
    # blend in the epilogue of a block's last 1x1 convolution (float32 and bf16; in bf16 with the blend's other input,
    # the old stage's from-audio layer, re-computed there), and the one rule of backward sweeps without weight
    # gradients: the data gradients of a block's first 3x3 convolution and of the from-audio layer as one launch.
    use_conv_chains = os.environ.get("MGX_CONV_CHAINS", "1") != "0"      # MGX_CONV_CHAINS=0: layer by layer (measurements)
    # True: the same groups of a TAPED bf16 pass run as one launch too (mgx_conv_chain_bf16_taped): every layer's
    # output is written for the backward, but the inner ones are not read back (256 instead of 384 bytes a position
    # and three-layer group); bit-identical tape.
    use_taped_chains = os.environ.get("MGX_TAPED_CHAINS", "1") != "0"

    def _conv_chain(self, plan, i, n_consumers, x):
        """The longest run of plan steps starting at i that mgx_conv_chain_bf16 serves as one launch, or None."""
        from .layers import conv_chain_eligible
        ops_ = [plan[i]]
        while len(ops_) < 3 and i + len(ops_) < len(plan):
            prev, nxt = ops_[-1], plan[i + len(ops_)]
            if (isinstance(nxt, _FusedOp) or nxt.act_node is None or len(nxt.inputs) != 1
                    or nxt.inputs[0] is not prev.output or n_consumers.get(id(prev.output), 0) != 1
                    or prev.output is self.output):
                break
            ops_.append(nxt)
        while len(ops_) >= 2:
            if conv_chain_eligible([o.node.layer for o in ops_], x):
                return ops_
            ops_.pop()
        return None

    def forward(self, x, save: bool = False):
        """Run the plan.  Returns (y, tape); tape is None unless save=True."""
        x = self._to_device(x)
        vals: Dict[int, torch.Tensor] = {id(self.input): x}
        tape = Tape() if save else None
        fused = Model.use_fused_groups and self._plan_infer is not self._plan
        if fused and save:
            fused = Model.use_fused_backward and all(op.group.backward is not None for op in self._plan_infer
                                                     if isinstance(op, _FusedOp))
        plan = self._plan_infer if fused else self._plan
        last_use = self._last_use if plan is self._plan else self._last_use_infer
        if save:
            tape.plan = plan
        producer, n_consumers = self._plan_links(plan)
        skip = -1
        # outputs of from-audio layers not computed yet (tape-free bf16 passes): id -> (layer, alpha, audio).  The
        # fade-in blend re-computes such an input in its epilogue; any other reader gets it computed on first use.
        deferred: Dict[int, tuple] = {}

        def value_of(t):
            pending = deferred.pop(id(t), None)
            if pending is not None:
                vals[id(t)] = pending[0].forward([pending[2]], False, act_alpha=pending[1])[0]
            return vals[id(t)]

        for i, op in enumerate(plan):
            if i <= skip:
                continue
            xs = [value_of(t) for t in op.inputs]
            if (not save and Model.use_conv_chains and not isinstance(op, _FusedOp) and op.act_node is not None
                    and K.bf16_enabled() and xs[0].dtype == torch.float32 and op.output is not self.output
                    and n_consumers.get(id(op.output), 0) == 1):
                # the old stage's from-audio convolution on the pooled audio, read only by the fade-in WeightedSum
                # as its first input: left to the blend's epilogue (mgx_conv1x1_blend_audio_bf16)
                from .layers import blend_audio_source
                reader = next((o for o in plan[i + 1:] if any(t is op.output for t in o.inputs)), None)
                if (reader is not None and not isinstance(reader, _FusedOp)
                        and getattr(reader.node.layer, "is_weighted_sum", False) and len(reader.inputs) == 2
                        and reader.inputs[0] is op.output and reader.inputs[1] is not op.output
                        and blend_audio_source(op.node.layer, xs[0])):
                    deferred[id(op.output)] = (op.node.layer, op.act_node.layer.alpha_value, xs[0])
                    for t in op.inputs:
                        if last_use.get(id(t)) == i and t is not self.output:
                            vals.pop(id(t), None)
                    continue
            if (not save and Model.use_conv_chains and not isinstance(op, _FusedOp) and op.act_node is not None
                    and K.bf16_enabled() and xs[0].dtype == torch.float32 and i + 1 < len(plan)):
                # the from-audio convolution and the 3x3 convolution behind it: the 128-channel tensor between
                # them is re-computed per tile instead of stored (mgx_audio_conv3_bf16; bit-identical results)
                nxt = plan[i + 1]
                if (not isinstance(nxt, _FusedOp) and nxt.act_node is not None and len(nxt.inputs) == 1
                        and nxt.inputs[0] is op.output and n_consumers.get(id(op.output), 0) == 1
                        and op.output is not self.output):
                    from .layers import audio_conv_eligible, audio_conv_forward
                    if audio_conv_eligible(op.node.layer, nxt.node.layer, xs[0]):
                        vals[id(nxt.output)] = audio_conv_forward(op.node.layer, op.act_node.layer.alpha_value,
                                                                  nxt.node.layer, nxt.act_node.layer.alpha_value, xs[0])
                        for t in op.inputs:
                            if last_use.get(id(t)) == i and t is not self.output:
                                vals.pop(id(t), None)
                        skip = i + 1
                        continue
            if (not save and Model.use_conv_chains and not isinstance(op, _FusedOp) and op.act_node is not None
                    and i + 1 < len(plan)):
                # the 1x1 convolution that ends a block and the fade-in WeightedSum behind it: the blend runs in the
                # convolution's epilogue (mgx_conv1x1_blend_bf16 / mgx_conv1x1_blend_f32; bit-identical results)
                nxt = plan[i + 1]
                if (not isinstance(nxt, _FusedOp) and getattr(nxt.node.layer, "is_weighted_sum", False)
                        and len(nxt.inputs) == 2 and nxt.inputs[1] is op.output and nxt.inputs[0] is not op.output
                        and n_consumers.get(id(op.output), 0) == 1 and op.output is not self.output
                        and (id(nxt.inputs[0]) in vals or id(nxt.inputs[0]) in deferred)):
                    from .layers import conv_blend_eligible, conv_blend_forward, conv_blend_audio_eligible, \
                        conv_blend_audio_forward
                    src = deferred.get(id(nxt.inputs[0]))
                    if src is not None and conv_blend_audio_eligible(op.node.layer, xs[0], src[2]):
                        del deferred[id(nxt.inputs[0])]
                        vals[id(nxt.output)] = conv_blend_audio_forward(op.node.layer, op.act_node.layer.alpha_value,
                                                                        xs[0], src[0], src[1], src[2],
                                                                        nxt.node.layer.alpha.value())
                        for j, o_ in ((i, op), (i + 1, nxt)):
                            for t in o_.inputs:
                                if last_use.get(id(t)) == j and t is not self.output:
                                    vals.pop(id(t), None)
                        skip = i + 1
                        continue
                    old = value_of(nxt.inputs[0])
                    if conv_blend_eligible(op.node.layer, xs[0], old):
                        vals[id(nxt.output)] = conv_blend_forward(op.node.layer, op.act_node.layer.alpha_value, xs[0],
                                                                  old, nxt.node.layer.alpha.value())
                        for j, o_ in ((i, op), (i + 1, nxt)):
                            for t in o_.inputs:
                                if last_use.get(id(t)) == j and t is not self.output:
                                    vals.pop(id(t), None)
                        skip = i + 1
                        continue
            if (Model.use_conv_chains and not isinstance(op, _FusedOp) and op.act_node is not None
                    and K.bf16_enabled() and xs[0].dtype == torch.bfloat16
                    and (not save or (Model.use_taped_chains and not K.BF16_SAVE_DERIVATIVE))):
                chain = self._conv_chain(plan, i, n_consumers, xs[0])
                if chain is not None:
                    from .layers import conv_chain_forward, conv_chain_forward_taped
                    chain_layers = [o.node.layer for o in chain]
                    chain_alphas = [o.act_node.layer.alpha_value for o in chain]
                    if save:
                        # one launch that also writes the inner outputs; the tape gets every layer's entry
                        vals[id(chain[-1].output)], entries = conv_chain_forward_taped(chain_layers, chain_alphas, xs[0])
                        tape.extend(entries)
                    else:
                        vals[id(chain[-1].output)] = conv_chain_forward(chain_layers, chain_alphas, xs[0])
                    for t in op.inputs:
                        if last_use.get(id(t)) == i and t is not self.output:
                            vals.pop(id(t), None)
                    skip = i + len(chain) - 1
                    continue
            # two consecutive convolution + activation steps whose intermediate nothing else reads run as one
            # launch when no tape is kept (bf16 path; the intermediate stays in LDS)
            if (not save and Model.use_conv_pairs and not isinstance(op, _FusedOp) and op.act_node is not None
                    and i + 1 < len(plan)):
                nxt = plan[i + 1]
                if (not isinstance(nxt, _FusedOp) and nxt.act_node is not None and len(nxt.inputs) == 1
                        and nxt.inputs[0] is op.output and n_consumers.get(id(op.output), 0) == 1
                        and op.output is not self.output):
                    from .layers import conv_pair_eligible, conv_pair_forward
                    if conv_pair_eligible(op.node.layer, nxt.node.layer, xs[0]):
                        vals[id(nxt.output)] = conv_pair_forward(op.node.layer, op.act_node.layer.alpha_value,
                                                                 nxt.node.layer, nxt.act_node.layer.alpha_value, xs[0])
                        for t in op.inputs:
                            if last_use.get(id(t)) == i and t is not self.output:
                                vals.pop(id(t), None)
                        skip = i + 1
                        continue
            if isinstance(op, _FusedOp):
                if save:
                    vals[id(op.output)], saved = op.group.run_save(xs[0])
                    tape.append(saved)
                else:
                    vals[id(op.output)] = op.group.run(xs[0])
                for t in op.inputs:
                    if last_use.get(id(t)) == i and t is not self.output:
                        vals.pop(id(t), None)
                continue
            if op.act_node is not None:
                y, saved = op.node.layer.forward(xs, save, act_alpha=op.act_node.layer.alpha_value)
            else:
                y, saved = op.node.layer.forward(xs, save)
            vals[id(op.output)] = y
            if save:
                tape.append(saved)
            for t in op.inputs:
                if last_use.get(id(t)) == i and t is not self.output:
                    vals.pop(id(t), None)
        return value_of(self.output), tape

    def __call__(self, x, training: bool = False):
        if isinstance(x, KTensor):
            raise TypeError("nested models are not supported")
        y, _ = self.forward(x, save=False)
        return y

    def predict(self, x, batch_size: Optional[int] = None):
        x = np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x
        n = x.shape[0]
        bs = batch_size or 32
        outs = []
        for i in range(0, n, bs):
            outs.append(self(x[i:i + bs]).cpu().numpy())
        return np.concatenate(outs, axis=0)

    def bind_grads(self, slot: int = 0) -> torch.Tensor:
        """Point every trainable variable's `.grad` at its slice of flat gradient buffer `slot` of
        this model (allocated once per slot, same layout) and return that buffer.  Slot 0 is the
        one the optimiser reads; slot 1 holds the real-batch pass of the discriminator step."""
        if self._grad_layout is None:
            layout, off = [], 0
            for v in self.trainable_variables:
                n = v.size
                layout.append((v, off, n))
                off += (n + 3) & ~3           # keep every slice 16-byte aligned
            self._grad_layout = layout
            self._grad_size = max(off, 4)
            self._grad_flats = {}
        flat = self._grad_flats.get(slot)
        if flat is None:
            flat = K.zeros((self._grad_size,))
            self._grad_flats[slot] = flat
            if slot == 0:
                self._grad_flat = flat
        for v, off, n in self._grad_layout:
            v.grad = flat[off:off + n]
        return flat

    def grad_layout(self):
        if self._grad_layout is None:
            self.bind_grads()
        return self._grad_layout

    def backward(self, tape, dy: torch.Tensor, need_input_grad: bool = False, need_weight_grads: bool = True):
        """Reverse sweep.  Weight gradients are ADDED into the variables' bound `.grad` views.
        Returns the gradient w.r.t. the model input (or None)."""
        grads: Dict[int, Tuple[torch.Tensor, bool]] = {id(self.output): (dy, False)}
        plan = getattr(tape, "plan", None) or self._plan
        producer, n_consumers = self._plan_links(plan)
        for i in range(len(plan) - 1, -1, -1):
            op = plan[i]
            g = grads.pop(id(op.output), None)
            if g is None:
                continue
            need = [(t is not self.input) or need_input_grad for t in op.inputs]
            existing: List[Optional[torch.Tensor]] = []
            pending = []
            for t in op.inputs:
                e = grads.get(id(t))
                if e is not None and not e[1]:
                    pending.append(e[0])      # aliased buffer: never accumulate into it
                    existing.append(None)
                else:
                    pending.append(None)
                    existing.append(e[0] if e is not None else None)
            if isinstance(op, _FusedOp):
                dx = op.group.backward(tape[i], g[0], need_weight_grads)
                if not need[0]:
                    res = [None]
                elif existing[0] is not None:
                    ops.add_inplace(existing[0], dx)
                    res = [(existing[0], True)]
                else:
                    res = [(dx, True)]
            elif (not need_weight_grads and Model.use_conv_chains and getattr(op.node.layer, "fusable_activation", False)
                  and self._audio_dgrad(plan, tape, i, producer, n_consumers, grads, g[0], need_input_grad)):
                continue
            elif getattr(op.node.layer, "fusable_activation", False):
                # the data-gradient kernel of a contraction can multiply its result by the saved
                # activation derivative of the layer that produced its input (single consumer only)
                j = producer.get(id(op.inputs[0]))
                outmul = tape[j] if (j is not None and n_consumers.get(id(op.inputs[0])) == 1
                                     and hasattr(tape[j], "premultiplied")) else None
                res = op.node.layer.backward(tape[i], g[0], need, existing, need_weight_grads, outmul=outmul)
            elif getattr(op.node.layer, "takes_outmuls", False):
                # WeightedSum: the same for each of its inputs
                oms = []
                for t in op.inputs:
                    j = producer.get(id(t))
                    oms.append(tape[j] if (j is not None and n_consumers.get(id(t)) == 1
                                           and hasattr(tape[j], "premultiplied")) else None)
                res = op.node.layer.backward(tape[i], g[0], need, existing, need_weight_grads, outmuls=oms)
            else:
                res = op.node.layer.backward(tape[i], g[0], need, existing, need_weight_grads)
            tape[i] = None
            for t, r, pend in zip(op.inputs, res, pending):
                if r is None:
                    continue
                if pend is not None:
                    out = K.empty(r[0].shape)
                    ops.add_n([pend, r[0]], out)
                    grads[id(t)] = (out, True)
                else:
                    grads[id(t)] = r
        gi = grads.get(id(self.input))
        return gi[0] if gi is not None else None

    def _audio_dgrad(self, plan, tape, i, producer, n_consumers, grads, dy, need_input_grad) -> bool:
        """Backward sweeps without weight gradients: the data gradient of a block's first 3x3 convolution and of the
        from-audio layer in front of it as one launch that never writes the 128-channel gradient between them.  True
        when it ran (the from-audio step then finds no gradient waiting and is skipped)."""
        from .layers import audio_dgrad_eligible, audio_dgrad_backward
        op = plan[i]
        j = producer.get(id(op.inputs[0]))
        if j is None or n_consumers.get(id(op.inputs[0])) != 1 or isinstance(plan[j], _FusedOp):
            return False
        prod = plan[j]
        if len(prod.inputs) != 1 or not audio_dgrad_eligible(op.node.layer, tape[i], prod.node.layer, tape[j]):
            return False
        t_in = prod.inputs[0]
        if t_in is self.input and not need_input_grad:
            return False
        e = grads.get(id(t_in))
        if e is not None and not e[1]:
            return False                      # an aliased buffer is waiting there: the layer-by-layer path adds out of place
        dxa = audio_dgrad_backward(op.node.layer, tape[i], prod.node.layer, tape[j], dy, e[0] if e is not None else None)
        grads[id(t_in)] = (dxa, True)
        tape[i] = None
        tape[j] = None
        return True

    # -- persistence (numpy archive; the reference's .h5 path needs h5py/Keras) --
    def get_weights_dict(self) -> Dict[str, np.ndarray]:
        out = {}
        for l in self.layers:
            for v in l.weights:
                out[v.name] = v.numpy()
        return out

    def set_weights_dict(self, weights: Dict[str, np.ndarray], strict: bool = True):
        for l in self.layers:
            for v in l.weights:
                if v.name in weights:
                    v.assign(np.asarray(weights[v.name]).reshape(v.shape))
                elif strict:
                    raise KeyError("no value for %s" % v.name)

    def save(self, path: str):
        np.savez(path, **{k.replace("/", "|"): v for k, v in self.get_weights_dict().items()})

    def load_weights(self, path: str):
        with np.load(path if path.endswith(".npz") else path + ".npz") as f:
            self.set_weights_dict({k.replace("|", "/"): f[k] for k in f.files})
