"""Command-line options for the convnet driver (Python 3).

What is kept from the reference's Python/options.py:79-178 is its behaviour and its names: an OptionsParser over getopt in
which an option is mandatory unless it has a default or another option excuses it, `requires` / `set_once` relations,
defaults given as expressions over the other options, and the merge of command-line values over the options saved in a
checkpoint.  The class names stay because checkpoints pickle the OptionsParser object under them
(Python/gpumodel.py:272-273); the messages stay because users and scripts read them.

How it is built here: a value type is a row of `_VALUE_TYPES` (name, usage string, converter, type test, formatter, complaint)
turned into a parser class by `_value_parser`; an Option carries its relations; the OptionsParser is a dict of Options plus
the three passes over it (command line, expression defaults, merge).
"""
import getopt as _getopt
import os.path
import re
import sys

_BOLD, _PLAIN = "\033[1m", "\033[0m"


class OptionException(Exception):
    """Anything wrong with an option's value or with the relations between options."""


class OptionMissingException(OptionException):
    """A mandatory option (or one another option requires) was not given: the caller prints the usage."""


class OptionExpression(object):
    """A default given as a Python expression over the other options' values,
    e.g. OptionExpression("[-1] * num_gpus") (Python/gpumodel.py:313)."""

    def __init__(self, expr):
        self.expr = expr

    def evaluate(self, env):
        try:
            return eval(self.expr, {}, dict(env))
        except Exception as problem:
            raise OptionException("expression '%s': unable to parse: %s" % (self.expr, problem))


# ---------------------------------------------------------------- value types
class OptionParser(object):
    """Base of the value types: `parse` (command-line string -> value), `to_string`, `is_type` (does an evaluated
    expression default have the right type), `get_type_str` (what the usage prints between <>)."""
    type_str = None
    parse = to_string = staticmethod(str)

    @classmethod
    def get_type_str(cls):
        return cls.type_str


def _value_parser(name, type_str, convert, accepts, show, complaint):
    def parse(text):
        try:
            return convert(text)
        except (TypeError, ValueError):
            raise OptionException(complaint)
    body = dict(type_str=type_str, parse=staticmethod(parse), to_string=staticmethod(show), is_type=staticmethod(accepts),
                __module__=__name__, __doc__="Option value of type <%s>." % type_str)
    return type(name, (OptionParser,), body)


def _zero_or_one(text):
    number = int(text)
    if number not in (0, 1):
        raise ValueError(text)
    return number


def _int_or_span(text):
    """`a-b` (inclusive) or a single integer -> list of ints (Python/options.py:328-336)."""
    span = re.fullmatch(r"(\d+)-(\d+)", text)
    if span is None:
        return [int(text)]
    first, last = map(int, span.groups())
    return list(range(first, last + 1))


def _exactly(kind, also=lambda value: True):
    return lambda value: type(value) is kind and also(value)


# name, usage string, converter, type test, formatter, complaint
_VALUE_TYPES = (
    ("StringOptionParser", "string", str, lambda value: isinstance(value, str), str, None),
    ("IntegerOptionParser", "int", int, _exactly(int), str, "argument is not an integer"),
    ("BooleanOptionParser", "0/1", _zero_or_one, _exactly(int, (0, 1).__contains__), str, "argument is not a boolean"),
    ("FloatOptionParser", "float", float, _exactly(float), "{:.6g}".format, "argument is not a float"),
    ("RangeOptionParser", "int[-int]", _int_or_span, lambda value: isinstance(value, list),
     lambda value: "%d-%d" % (value[0], value[-1]), "argument is neither an integer nor a range"),
)
for _row in _VALUE_TYPES:
    globals()[_row[0]] = _value_parser(*_row)
del _row


class ListOptionParser(OptionParser):
    """Separator-delimited list; `parsers` is one parser (any length) or a list of parsers (fixed form)."""

    def __init__(self, parsers, sepchar=","):
        self.parsers, self.sepchar = parsers, sepchar

    @property
    def _fixed(self):
        return isinstance(self.parsers, list)

    def _each(self, items):
        """(parser, item) pairs."""
        return zip(self.parsers, items) if self._fixed else ((self.parsers, item) for item in items)

    def parse(self, value):
        items = value.split(self.sepchar)
        if self._fixed and len(items) != len(self.parsers):
            raise OptionException("requires %d arguments, given %d" % (len(self.parsers), len(items)))
        try:
            return [parser.parse(item) for parser, item in self._each(items)]
        except OptionException:
            raise OptionException("argument is not of the form %s" % self.get_type_str())

    def to_string(self, value):
        return self.sepchar.join(parser.to_string(item) for parser, item in self._each(value))

    def get_type_str(self):
        if self._fixed:
            return self.sepchar.join(parser.get_type_str() for parser in self.parsers)
        return "%s%s..." % (self.parsers.get_type_str(), self.sepchar)

    is_type = staticmethod(lambda value: isinstance(value, list))


# ---------------------------------------------------------------- one option
class Option(object):
    def __init__(self, letter, name, desc, parser, set_once, default, excuses, requires, save):
        if name is None:
            raise ValueError("an option needs a name")
        vars(self).update(letter=letter, name=name, desc=desc, parser=parser, set_once=set_once, default=default, excuses=excuses,
                          requires=requires, save=save, value=None, value_given=False)
        # single-letter options take one dash, the others two (Python/options.py:49)
        self.prefixed_letter = ("-" if len(letter) == 1 else "--") + letter

    def __str__(self):
        return "%s (%s)" % (self.prefixed_letter, self.desc)

    @property
    def mandatory(self):
        return self.default is None

    @property
    def defaulted(self):
        return not (self.value_given or self.mandatory)

    def excuses_option(self, other):
        return self.excuses == OptionsParser.EXCLUDE_ALL or other.name in self.excuses

    def set_value(self, value, parse=True):
        try:
            self.value, self.value_given = (self.parser.parse(value) if parse else value), True
        except OptionException as problem:
            raise OptionException("Unable to parse option %s: %s" % (self, problem))

    def set_default(self):
        if not self.mandatory:
            self.value = self.default

    def eval_expr_default(self, env):
        if self.value_given or not isinstance(self.default, OptionExpression):
            return
        try:
            result = self.value = self.default.evaluate(env)
            if not self.parser.is_type(result):
                raise OptionException("expression result %s is not of right type (%s)" % (result, self.parser.get_type_str()))
        except Exception as problem:
            raise OptionException("Unable to set default value for option %s: %s" % (self, problem))

    def get_str_value(self, get_default_str=False):
        shown = self.default if get_default_str else self.value
        if shown is None:
            return ""
        return shown.expr if isinstance(shown, OptionExpression) else self.parser.to_string(shown)


# ---------------------------------------------------------------- the option set
class OptionsParser(object):
    SORT_LETTER, SORT_DESC, SORT_EXPR_LAST = 1, 2, 3
    EXCLUDE_ALL = "all"
    _SORT_KEYS = {SORT_LETTER: lambda o: o.letter, SORT_DESC: lambda o: o.desc,
                  SORT_EXPR_LAST: lambda o: isinstance(o.default, OptionExpression)}

    def __init__(self):
        self.options = dict()

    def add_option(self, letter, name, parser, desc, set_once=False, default=None, excuses=(), requires=(), save=True):
        if name in self.options:
            raise ValueError("option '%s' defined twice" % name)
        self.options[name] = Option(letter, name, desc, parser, set_once, default, excuses, requires, save)

    def set_value(self, name, value, parse=True):
        self.options[name].set_value(value, parse=parse)

    def get_value(self, name):
        return self.options[name].value

    def delete_option(self, name):
        self.options.pop(name, None)

    def _getopt_spec(self):
        flags = [o.letter for o in self.options.values()]
        return "".join(f + ":" for f in flags if len(f) == 1), [f + "=" for f in flags if len(f) > 1]

    def parse(self, eval_expr_defaults=False, argv=None):
        """Values from the command line (sys.argv when `argv` is None), defaults for the rest; raises
        OptionMissingException for a mandatory option that nothing on the command line excuses."""
        shorts, longs = self._getopt_spec()
        given = dict(_getopt.getopt(sys.argv[1:] if argv is None else argv, shorts, longopts=longs)[0])
        on_line = lambda option: option.prefixed_letter in given
        by_letter = self.get_options_list(sort_order=self.SORT_LETTER)
        for o in self.get_options_list(sort_order=self.SORT_EXPR_LAST):
            if on_line(o):
                o.set_value(given[o.prefixed_letter])
                lacking = [need for need in by_letter if need.name in o.requires and not on_line(need)]
                if lacking:
                    raise OptionMissingException("Option %s requires option %s" % (o, lacking[0]))
                continue
            if o.mandatory and not any(on_line(other) and other.excuses_option(o) for other in self.options.values()):
                raise OptionMissingException("Option %s not supplied" % o)
            o.set_default()
        if eval_expr_defaults:
            self.eval_expr_defaults()
        return self.options

    def merge_from(self, op2):
        """Take op2's explicitly given (or non-saved) options over this (loaded) set; options this
        set lacks are added (Python/options.py:160-170)."""
        for name, theirs in op2.options.items():
            if name not in self.options:
                self.options[name] = theirs
            elif theirs.value_given or not theirs.save:
                if theirs.set_once:
                    raise OptionException("Option %s cannot be changed" % theirs)
                self.options[name] = theirs

    def eval_expr_defaults(self):
        env = {name: o.value for name, o in self.options.items()}
        for o in self.options.values():
            o.eval_expr_default(env)

    def all_values_given(self):
        return any(o.value_given for o in self.options.values() if not o.mandatory)

    def get_options_list(self, sort_order=SORT_LETTER):
        return sorted(self.options.values(), key=self._SORT_KEYS.get(sort_order, self._SORT_KEYS[self.SORT_DESC]))

    def print_usage(self, print_constraints=False):
        print("%s usage:" % os.path.basename(sys.argv[0]))
        everyone = self.options.values()
        flags_of = lambda chosen: ", ".join(sorted(o.prefixed_letter for o in chosen))
        rows = []
        for o in self.get_options_list():
            rows.append((self._bold(o.prefixed_letter) + " <%s>" % o.parser.get_type_str(), o.desc,
                         None if o.mandatory else "[%s]" % o.get_str_value(get_default_str=True),
                         flags_of(e for e in everyone if e.excuses_option(o)) if o.mandatory else " ",
                         flags_of(r for r in everyone if r.name in o.requires)))
        rows.sort(key=lambda row: row[2] is None)           # optional ones first, the mandatory ones after (stable)
        widths = [self._longest_value(rows, key=lambda row, i=i: row[i]) for i in range(4)]
        heads = ["    Option", "Description", "Default"] + ["Excused by", "Requires"] * bool(print_constraints)
        print(" ".join(self._bold(head.ljust(width)) for head, width in zip(heads, widths + [0])))
        for flag, desc, default, excused_by, needs in rows:
            lead = ("     %s  -" if default is None else "    [%s] -") % flag.ljust(widths[0])
            tail = " %s %s" % (excused_by.ljust(widths[3]), needs) if print_constraints else ""
            print("%s %s %s%s" % (lead, desc.ljust(widths[1]), (default or " ").ljust(widths[2]), tail))

    def print_values(self):
        desc_width = self._longest_value(self.options.values(), key=lambda o: o.desc)
        value_width = self._longest_value([o for o in self.options.values() if o.defaulted], key=Option.get_str_value)
        for o in self.get_options_list(sort_order=self.SORT_DESC):
            print("%s: %s %s" % (o.desc.ljust(desc_width), o.get_str_value().ljust(value_width), "[DEFAULT]" if o.defaulted else ""))

    @staticmethod
    def _longest_value(values, key=lambda x: x):
        return max((len(key(v)) for v in values if key(v) is not None), default=0)

    @staticmethod
    def _bold(text):
        return "".join((_BOLD, text, _PLAIN))
